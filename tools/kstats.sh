#!/bin/bash
# per-kernel times of the hot path, one context (run on the GPU box through gpurun): tools/kstats.sh <tag> [extra bench flags]
TAG=${1:-kstats}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
B="python3 $R/bench.py --no-cpu-baseline --no-extras --blocks 1"
$B --steps 1 --warmup 0 > $OUT/warm.json 2> $OUT/warm.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats1 -o stats1 -- $B --steps 6 --warmup 2 --contexts 1 "$@" > $OUT/stats1.json 2> $OUT/stats1.err
f=$(find $OUT/stats1 -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv
python3 - $OUT/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-60s calls %5s  avg %10.1f us  total %8.2f ms  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
tail -1 $OUT/stats1.json | cut -c1-330
