#!/bin/bash
# PMC passes for the hot-path kernels (run on the GPU box through gpurun; counters in their own runs).
# usage: tools/pmc_pass.sh <tag> [reads]
TAG=${1:-pmc}; READS=${2:-4096}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
python3 $R/bench.py --reads-per-gpu $READS --steps 1 --warmup 0 --no-cpu-baseline > $OUT/warm.json 2> $OUT/warm.err   # builds the index cache
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o stats -- python3 $R/bench.py --reads-per-gpu $READS --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats.json 2> $OUT/stats.err
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/sq -o sq -- python3 $R/bench.py --reads-per-gpu $READS --steps 1 --warmup 0 --no-cpu-baseline > $OUT/sq.json 2> $OUT/sq.err
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python3 $R/bench.py --reads-per-gpu $READS --steps 1 --warmup 0 --no-cpu-baseline > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write -o write -- python3 $R/bench.py --reads-per-gpu $READS --steps 1 --warmup 0 --no-cpu-baseline > $OUT/write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -30
python3 $R/tools/pmc_summary.py $OUT --emit $OUT/pmc_latest.json --kernel k_ext_rows --reads $READS > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
