#!/bin/bash
# two counter passes (SQ activity, FETCH/WRITE) of a one-context bench step: tools/pmc_quick.sh <tag>   (environment switches pass through)
TAG=${1:-pmcq}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
B="python3 $R/bench.py --no-cpu-baseline --no-extras --blocks 1"
$B --steps 1 --warmup 0 > $OUT/warm.json 2> $OUT/warm.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats1 -o stats1 -- $B --steps 3 --warmup 1 --contexts 1 > $OUT/stats1.json 2> $OUT/stats1.err
pass() { name=$1; shift; timeout 240 rocprofv3 --output-format csv --kernel-trace --pmc "$@" -d $OUT/$name -o $name -- $B --steps 1 --warmup 0 --contexts 1 > $OUT/$name.json 2> $OUT/$name.err || echo "pass $name failed"; }
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
