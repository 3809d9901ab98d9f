#!/bin/bash
# the N > 1 path of bench.py on a 1-GPU box: two ranks share GPU 0, torch.distributed over gloo; small genome, one context per rank (memory)
cd $GRAFT_REPO_ROOT
YAHA_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 4 --warmup 1 --genome-mbp 100 --contexts 1 --e2e-reads 65536 > gpurun_out/r03_n2_dryrun.json 2> gpurun_out/r03_n2_dryrun.err; echo "rc $?"
tail -5 gpurun_out/r03_n2_dryrun.err | cut -c1-300; python - <<'PY'
import json; j=json.loads(open('gpurun_out/r03_n2_dryrun.json').read().strip().split("\n")[-1])
print({k: j.get(k) for k in ('n_gpus','value','ms_per_step','e2e_reads_per_s','steady_reads_per_s')}); print(j.get('end_to_end'))
PY
