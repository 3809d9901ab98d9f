#!/bin/bash
# G-hg18scale: a 3.1 Gbp synthetic genome with hg18-like sequence ratios and repeat content, indexed with the reference's defaults
# (-L 15: 4.3 GB table + ~12 GB ROA) on the GPU, then the bench step on it: does reads/s depend on the size of the index?  Run on the GPU box.
R=$GRAFT_REPO_ROOT; export YAHA_TIMING=1
cd $R
python3 bench.py --genome-mbp 3100 --steps 4 --warmup 1 --no-extras --cpu-seconds 24 > gpurun_out/hg18scale_bench.json 2> gpurun_out/hg18scale_bench.err
grep -E "\[yaha\]|\[bench\]|Maximum resident|Elapsed|hits|error|Error" gpurun_out/hg18scale_bench.err | head -40
ls -la /tmp/yaha_bench_cache/ | head
python3 - <<PY
import json
d=json.loads(open("gpurun_out/hg18scale_bench.json").read().strip().split("\n")[-1])
print({k: d[k] for k in ("value","ms_per_step")}, d["per_read"], d.get("cpu_baseline"), d["stage_ms_per_step"])
PY
