#!/bin/bash
# one SQ counter pass + one-context kernel stats of the bench with the built library: the per-kernel table (vector instructions, wait shares): $1 = tag
TAG=${1:-pmcsq}; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
B="python3 $R/bench.py --no-cpu-baseline --no-extras --blocks 1"
$B --steps 1 --warmup 0 > $OUT/warm.json 2> $OUT/warm.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats1 -o stats1 -- $B --steps 6 --warmup 2 --contexts 1 > $OUT/stats1.json 2> $OUT/stats1.err
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/sq -o sq -- $B --steps 1 --warmup 0 --contexts 1 > $OUT/sq.json 2> $OUT/sq.err
python3 $R/tools/pmc_table.py $OUT > $OUT/table.txt 2>&1; cut -c1-130 $OUT/table.txt | head -44
