#!/bin/bash
# per-kernel times of one context incl. the post-filter stage (the bench's extras leg runs it): rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r03_oqc_prof}; mkdir -p $O
python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/warm.json 2> $O/warm.err
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o stats -- python3 $R/bench.py --steps 4 --warmup 1 --contexts 1 --no-cpu-baseline --e2e-reads 16384 > $O/bench.json 2> $O/bench.err
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); head -40 $f | cut -c1-160
