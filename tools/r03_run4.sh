#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
python -m pytest tests -m gpu -x -q -k "postfilter or cli_drop_in or long_reads or config or bench_scale" > $O/r03_gputests4.log 2>&1; echo "pytest rc $?" >> $O/r03_gputests4.log; tail -6 $O/r03_gputests4.log
python bench.py > $O/r03_bench4.json 2> $O/r03_bench4.err; echo "bench rc $?"; python - <<'PY'
import json; j=json.load(open('gpurun_out/r03_bench4.json'))
print({k: j.get(k) for k in ('value','ms_per_step','e2e_reads_per_s','steady_reads_per_s','value_int32','value_with_d2h','value_with_postfilter')}); print(j.get('d2h')); print(j['end_to_end']['cli_stats'])
PY
bash tools/r03_oqc_prof.sh r03_oqc_prof2 | grep -E "oqc|ext_rows_pk<false>" | cut -c1-150
