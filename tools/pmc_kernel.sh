#!/bin/bash
# SQ and LDS counters of the kernels whose name contains $1 (default k_seg_sort), one context, one step: two --pmc passes with --kernel-trace only.  tools/pmc_kernel.sh [substring] [tag]
export KSUB=${1:-k_seg_sort}; TAG=${2:-pmc_kernel}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; export OUT; mkdir -p $OUT
B="python3 $R/bench.py --no-cpu-baseline --no-extras"
$B --steps 1 --warmup 0 > $OUT/warm.json 2> $OUT/warm.err
pass() { name=$1; shift; timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc "$@" -d $OUT/$name -o $name -- $B --steps 1 --warmup 0 --contexts 1 > $OUT/$name.json 2> $OUT/$name.err || echo "pass $name failed"; }
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ["OUT"]
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(O+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if os.environ["KSUB"] not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"] or 0)
for k,v in sorted(agg.items()):
    w=v.get("SQ_WAVE_CYCLES",0) or 1
    print(k[:34], "VALU %.0f M SALU %.0f M wavecyc %.0f M VALUact %.1f%% wait_any %.1f%% wait_inst %.1f%% LDSinsts %.1f M LDSactive %.0f M conflict/active %.1f%% waitLDS %.1f%% vmemrd %.2f M" % (v.get("SQ_INSTS_VALU",0)/1e6, v.get("SQ_INSTS_SALU",0)/1e6, w/1e6, 100*v.get("SQ_ACTIVE_INST_VALU",0)/w, 100*v.get("SQ_WAIT_ANY",0)/w, 100*v.get("SQ_WAIT_INST_ANY",0)/w, v.get("SQ_INSTS_LDS",0)/1e6, v.get("SQ_LDS_IDX_ACTIVE",0)/1e6, 100*v.get("SQ_LDS_BANK_CONFLICT",0)/max(1,v.get("SQ_LDS_IDX_ACTIVE",0)), 100*v.get("SQ_WAIT_INST_LDS",0)/w, v.get("SQ_INSTS_VMEM_RD",0)/1e6))
PY
