#!/bin/bash
# counters of the post-filter kernels (one context; the bench's extras leg runs the stage): separate --pmc passes with --kernel-trace only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_oqc_pmc; mkdir -p $O
B="python3 $R/bench.py --steps 1 --warmup 0 --contexts 1 --no-cpu-baseline --e2e-reads 16384"
$B > $O/warm.json 2> $O/warm.err
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU -d $O/sq -o sq -- $B > $O/sq.json 2> $O/sq.err
timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM -d $O/lds -o lds -- $B > $O/lds.json 2> $O/lds.err
python3 - <<'PY'
import csv, glob, os, collections
O=os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03_oqc_pmc")
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
for f in glob.glob(O+"/**/*counter_collection.csv", recursive=True):
    seen=set()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if not k.startswith("k_oqc"): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"] or 0)
        if (k, r["Dispatch_Id"], os.path.basename(f)) not in seen and r["Counter_Name"] in ("SQ_WAVE_CYCLES",): n[k]+=1
        seen.add((k, r["Dispatch_Id"], os.path.basename(f)))
for k,v in agg.items():
    w=v.get("SQ_WAVE_CYCLES",0) or 1
    print(k, "dispatches", n[k], "VALU insts %.1f M, SALU %.1f M, wave cycles %.0f M, VALU active %.1f%%, waiting for memory or LDS %.1f%%, waiting to issue %.1f%%, LDS insts %.1f M, LDS conflict cycles / LDS active %.1f%%, VMEM rd %.2f M wr %.2f M"
          % (v.get("SQ_INSTS_VALU",0)/1e6, v.get("SQ_INSTS_SALU",0)/1e6, w/1e6, 100*v.get("SQ_ACTIVE_INST_VALU",0)/w, 100*v.get("SQ_WAIT_ANY",0)/w, 100*v.get("SQ_WAIT_INST_ANY",0)/w, v.get("SQ_INSTS_LDS",0)/1e6,
             100*v.get("SQ_LDS_BANK_CONFLICT",0)/max(1,v.get("SQ_LDS_IDX_ACTIVE",0)), v.get("SQ_INSTS_VMEM_RD",0)/1e6, v.get("SQ_INSTS_VMEM_WR",0)/1e6))
PY
