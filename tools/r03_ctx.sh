#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
for c in 3 4 5 3 4; do
  YGPU_STATS=1 python bench.py --steps 12 --warmup 2 --contexts $c --no-cpu-baseline --no-extras 2> /tmp/err.txt | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('contexts $c', round(j['value']), round(j['ms_per_step'],2))"
  grep -c "ranges [2-9]" /tmp/err.txt; grep "run:" /tmp/err.txt | tail -1 | cut -c1-200
done
