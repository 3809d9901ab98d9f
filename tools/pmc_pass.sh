#!/bin/bash
# PMC passes for the hot-path kernels (run on the GPU box through gpurun; counters in their own runs with --kernel-trace only).
# usage: tools/pmc_pass.sh <tag> [reads]
TAG=${1:-pmc}; READS=${2:-16384}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
B="python3 $R/bench.py --reads-per-gpu $READS --no-cpu-baseline --no-extras --blocks 1"
$B --steps 1 --warmup 0 > $OUT/warm.json 2> $OUT/warm.err   # builds the index cache
rocprofv3 -L > $OUT/counters_available.txt 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o stats -- $B --steps 6 --warmup 2 > $OUT/stats.json 2> $OUT/stats.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats1 -o stats1 -- $B --steps 6 --warmup 2 --contexts 1 > $OUT/stats1.json 2> $OUT/stats1.err
pass() { name=$1; shift; timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc "$@" -d $OUT/$name -o $name -- $B --steps 1 --warmup 0 --contexts 1 > $OUT/$name.json 2> $OUT/$name.err || echo "pass $name failed (see $name.err)"; }
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
pass fetch FETCH_SIZE
# (a pass with the TA_* counters hung the profiler on this pool until the call's limit: 20 minutes of GPU time; the TCC/SQ-wait passes are left out with it)
pass write WRITE_SIZE
# what FETCH_SIZE means for one-byte-per-lane streams (k_ext_rows' access shape): tools/micro/fetch_calib.hip
if [ -x $R/tools/micro/fetch_calib ]; then
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/calib -o calib -- $R/tools/micro/fetch_calib > $OUT/calib.json 2> $OUT/calib.err
fi
find $OUT -name "*.csv" | head -40
python3 $R/tools/pmc_summary.py $OUT --emit $OUT/pmc_latest.json --kernel k_ext_rows --reads $READS > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
