#!/bin/bash
# rocprofv3 kernel-trace summary of one bench run (run on the GPU box through gpurun).  usage: tools/stats_pass.sh <tag> [reads]
TAG=${1:-stats}; READS=${2:-8192}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
python3 $R/bench.py --reads-per-gpu $READS --steps 1 --warmup 0 --no-cpu-baseline > $OUT/warm.json 2> $OUT/warm.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o stats -- python3 $R/bench.py --reads-per-gpu $READS --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats.json 2> $OUT/stats.err
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/stats/stats_kernel_stats.csv")))
for r in rows[:22]:
    print("%-70s calls %4s avg %10.1f us  total %9.2f ms  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
cat $OUT/stats.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
