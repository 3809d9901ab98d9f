#!/bin/bash
# counter passes (SQ, LDS) of the bench with one build of the library, the k_seg_sort rows of the table: $1 = tag, $2 = library file in yaha_amd/csrc
TAG=${1:-pmcsort}; R=$GRAFT_REPO_ROOT; export YAHA_HIP_LIB=$R/yaha_amd/csrc/${2:-libyaha_hip.so}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
B="python3 $R/bench.py --no-cpu-baseline --no-extras --blocks 1"
$B --steps 1 --warmup 0 > $OUT/warm.json 2> $OUT/warm.err
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats1 -o stats1 -- $B --steps 6 --warmup 2 --contexts 1 > $OUT/stats1.json 2> $OUT/stats1.err
pass() { name=$1; shift; timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc "$@" -d $OUT/$name -o $name -- $B --steps 1 --warmup 0 --contexts 1 > $OUT/$name.json 2> $OUT/$name.err || echo "pass $name failed"; }
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
python3 $R/tools/pmc_table.py $OUT > $OUT/table.txt 2>&1; head -4 $OUT/table.txt | cut -c1-200; grep seg_sort $OUT/table.txt
python3 - $OUT <<'PY'
import csv, glob, sys, collections
c = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_seg_sort" in r["Kernel_Name"]: c[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"] or 0)
for k in sorted(c):
    v = c[k]; print(k, {n: round(x / 1e6, 1) for n, x in sorted(v.items())})
PY
