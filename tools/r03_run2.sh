#!/bin/bash
# round 3, second GPU call: new GPU tests (wide bands, bench-scale index vs the reference), bench with the direct collect + fast exit, host ceiling, at-scale parity logs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
python -m pytest tests -m gpu -x -q -k "wider or bench_scale or cli_drop_in or two_contexts or async" > $O/r03_gputests2.log 2>&1; echo "pytest rc $?" >> $O/r03_gputests2.log; tail -5 $O/r03_gputests2.log
python bench.py > $O/r03_bench2.json 2> $O/r03_bench2.err; echo "bench rc $?"; python - <<'PY'
import json; j=json.load(open('gpurun_out/r03_bench2.json'))
print({k: j.get(k) for k in ('value','ms_per_step','e2e_reads_per_s','steady_reads_per_s','value_int32','value_with_d2h')}); print(j['end_to_end'])
PY
python tools/host_ceiling.py > $O/r03_host_ceiling2.jsonl 2> $O/r03_host_ceiling2.err; tail -12 $O/r03_host_ceiling2.jsonl; tail -3 $O/r03_host_ceiling2.err
C=/tmp/yaha_bench_cache; X=$C/g3100m_s42.X15_01_65525S; R=$(ls $C/e2e_n1048576_l1000_s3000.fa)
YAHA_TIMING=1 yaha_amd/csrc/yaha -x $X -q $R -osh /dev/shm/o.sam 2> $O/r03_cli_timing2.txt; grep -v ticket $O/r03_cli_timing2.txt | tail -4; grep ticket $O/r03_cli_timing2.txt | sed -n '10,14p'
rm -f /dev/shm/o.sam
YAHA_PARITY_GENOME=g3100m_s42 python tools/big_parity.py 16384 > $O/r03_at_scale_validation_vs_reference_3100Mbp.log 2>&1; cat $O/r03_at_scale_validation_vs_reference_3100Mbp.log
YAHA_PARITY_GENOME=g3100m_s42 python tools/big_parity_opts.py 4000 > $O/r03_at_scale_validation_option_sets_3100Mbp.log 2>&1; tail -20 $O/r03_at_scale_validation_option_sets_3100Mbp.log
