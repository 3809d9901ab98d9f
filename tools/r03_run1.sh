#!/bin/bash
# round 3, first GPU call: regression of the GPU tier, the default bench, the host ceiling and the command line's timeline on the bench index
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r03_gputests1.log 2>&1; echo "pytest rc $?" >> $O/r03_gputests1.log; tail -3 $O/r03_gputests1.log
python bench.py > $O/r03_bench1.json 2> $O/r03_bench1.err; echo "bench rc $?"; tail -c 600 $O/r03_bench1.json
python tools/host_ceiling.py > $O/r03_host_ceiling1.jsonl 2> $O/r03_host_ceiling1.err; tail -12 $O/r03_host_ceiling1.jsonl
C=/tmp/yaha_bench_cache; X=$C/g3100m_s42.X15_01_65525S; R=$(ls $C/e2e_n1048576_l1000_s3000.fa)
for opts in "" "-ctx 2" "-ctx 4"; do
  YAHA_TIMING=1 yaha_amd/csrc/yaha -x $X -q $R -osh /dev/shm/o.sam $opts 2> $O/r03_cli_timing_$(echo $opts | tr -d ' -').txt
  echo "== cli [$opts]"; grep -v ticket $O/r03_cli_timing_$(echo $opts | tr -d ' -').txt | tail -4
done
rm -f /dev/shm/o.sam
