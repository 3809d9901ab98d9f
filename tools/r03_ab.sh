#!/bin/bash
# A/B of an environment switch on the default bench (three contexts, 12 steps), alternating: tools/r03_ab.sh VAR A B
cd $GRAFT_REPO_ROOT; V=$1; A=$2; B=$3
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
for rep in 1 2 3 4; do for x in $A $B; do
  env $V=$x python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$V=$x', round(j['value']), round(j['ms_per_step'],2))"
done; done
for x in $A $B; do
  env $V=$x python bench.py --steps 6 --warmup 2 --contexts 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('one context $V=$x', round(j['value']), round(j['ms_per_step'],2))"
done
