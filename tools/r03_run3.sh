#!/bin/bash
# round 3, third GPU call: the whole GPU tier (device post-filter, -S > 1 index on the device, wide bands, bench-scale index), bench, host ceiling, command-line timeline, at-scale parity
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r03_gputests3.log 2>&1; echo "pytest rc $?" >> $O/r03_gputests3.log; tail -15 $O/r03_gputests3.log
python bench.py > $O/r03_bench3.json 2> $O/r03_bench3.err; echo "bench rc $?"; python - <<'PY'
import json; j=json.load(open('gpurun_out/r03_bench3.json'))
print({k: j.get(k) for k in ('value','ms_per_step','e2e_reads_per_s','steady_reads_per_s','value_int32','value_with_d2h','value_with_postfilter')}); print(j.get('d2h')); print(j['end_to_end'])
PY
python tools/host_ceiling.py > $O/r03_host_ceiling3.jsonl 2> $O/r03_host_ceiling3.err; tail -16 $O/r03_host_ceiling3.jsonl; tail -3 $O/r03_host_ceiling3.err
C=/tmp/yaha_bench_cache; X=$C/g3100m_s42.X15_01_65525S; R=$(ls $C/e2e_n1048576_l1000_s3000.fa)
for mode in dev host; do
  if [ $mode = host ]; then export YAHA_HOST_OQC=1; else unset YAHA_HOST_OQC; fi
  YAHA_TIMING=1 yaha_amd/csrc/yaha -x $X -q $R -osh /dev/shm/o_$mode.sam 2> $O/r03_cli_timing3_$mode.txt; echo "== $mode filter"; grep -v ticket $O/r03_cli_timing3_$mode.txt | tail -3; grep ticket $O/r03_cli_timing3_$mode.txt | sed -n '10,12p'
done
unset YAHA_HOST_OQC
cmp <(grep -v "^@PG" /dev/shm/o_dev.sam) <(grep -v "^@PG" /dev/shm/o_host.sam) && echo "device-filtered and host-filtered SAM of 1 M reads: identical (minus @PG)"; rm -f /dev/shm/o_dev.sam /dev/shm/o_host.sam
YAHA_PARITY_GENOME=g3100m_s42 python tools/big_parity.py 16384 > $O/r03_at_scale_validation_vs_reference_3100Mbp.log 2>&1; cat $O/r03_at_scale_validation_vs_reference_3100Mbp.log
YAHA_PARITY_GENOME=g3100m_s42 python tools/big_parity_opts.py 4000 > $O/r03_at_scale_validation_option_sets_3100Mbp.log 2>&1; tail -20 $O/r03_at_scale_validation_option_sets_3100Mbp.log
