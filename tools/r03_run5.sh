#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r03_gputests5.log 2>&1; echo "pytest rc $?" >> $O/r03_gputests5.log; tail -6 $O/r03_gputests5.log
for d in 1 0; do
YGPU_SORT_DROP=$d python bench.py --no-cpu-baseline --e2e-reads 16384 > $O/r03_bench5_drop$d.json 2> $O/r03_bench5_drop$d.err; python - <<PY
import json; j=json.load(open('gpurun_out/r03_bench5_drop$d.json'))
print("drop=$d", {k: j.get(k) for k in ('value','ms_per_step','value_int32','value_with_postfilter')}); print({k: round(v,2) for k,v in j['stage_ms_per_step'].items()})
PY
done
YGPU_TRACE=1 python bench.py --steps 1 --warmup 1 --contexts 1 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep -E "hit sort" | tail -4
