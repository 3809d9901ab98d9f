#!/bin/bash
# One measurement of a hot-path change: the stage tests, the default bench three times (three contexts), one context, and the per-kernel times of one context.
# tools/r03_step.sh TAG [pytest -k expression]
cd $GRAFT_REPO_ROOT; TAG=${1:-step}; K=${2:-"seed or sort or stage or golden or bench_scale or chain"}
python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -3
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
for rep in 1 2 3; do
  python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('default contexts', round(j['value']), round(j['ms_per_step'],2))"
done
python bench.py --steps 6 --warmup 2 --contexts 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('one context', round(j['value']), round(j['ms_per_step'],2))"
cd /tmp && export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03_$TAG; mkdir -p $O
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --contexts 1 --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
python3 - "$O" <<'PY'
import csv, sys, os
f = os.path.join(sys.argv[1], "stats/stats_kernel_stats.csv")
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("kernel time per step (8 timed + 2 warm-up + the first batch's passes): %.2f ms / 10" % (tot / 1e6))
for r in rows[:40]: print("%-70s %5d %9.3f" % (r['Name'][:70], int(r['Calls']), float(r['TotalDurationNs']) / 1e6 / max(1, int(r['Calls']))))
PY
cp $O/stats/stats_kernel_stats.csv $O/kernel_stats.csv; rm -rf $O/stats
