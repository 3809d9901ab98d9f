#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "postfilter or cli_drop_in or long_reads or bench_scale" 2>&1 | tail -2
YGPU_OQC_PROF=1 python bench.py --steps 2 --warmup 1 --contexts 1 --no-cpu-baseline --e2e-reads 16384 2>&1 >/dev/null | grep "post-filter class" | tail -4
cd /tmp && export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r03_oqc_prof3; mkdir -p $O
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --contexts 1 --no-cpu-baseline --e2e-reads 16384 > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import csv, os
f = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r03_oqc_prof3/stats/stats_kernel_trace.csv")
rows=[r for r in csv.DictReader(open(f)) if 'k_oqc' in r['Kernel_Name']]
t0=int(rows[0]['Start_Timestamp'])
for r in rows[:11]: print(r['Kernel_Name'][:12], int(r['Grid_Size_X'])//64, r['LDS_Block_Size'], round((int(r['Start_Timestamp'])-t0)/1e6,3), round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6,3), r.get('Stream_Id'))
PY
