#!/bin/bash
# three-way (or more) A/B of environment settings on the default bench, alternating: tools/ab3.sh REPS "VAR=a" "VAR=b VAR2=c" ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
B="python bench.py --no-cpu-baseline --no-extras --steps 12 --warmup 2"
N=$1; shift
$B --steps 2 --warmup 1 > /dev/null 2>&1
for rep in $(seq 1 $N); do for s in "$@"; do env $s $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('[$s]', round(d['value']), round(d['ms_per_step'],2), d['ms_per_step_blocks'])"; done; done
