#!/bin/bash
# the command line on 10 kbp reads (BASELINE config 3's shape): device vs host post-filter, format time per batch
cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
C=/tmp/yaha_bench_cache; X=$C/g3100m_s42.X15_01_65525S; R=$C/cli10k.fa
[ -f $R ] || tools/yaha_sim reads --genome $C/g3100m_s42.fa --out $R --seed 77 --n 32768 --len 10000 --div 0.034
yaha_amd/csrc/yaha -x $X -q $R -osh /dev/shm/o.sam 2>/dev/null
for mode in dev host dev1000; do
  sleep 25; unset YAHA_HOST_OQC YGPU_OQC_MAX
  [ $mode = host ] && export YAHA_HOST_OQC=1
  YAHA_TIMING=1 yaha_amd/csrc/yaha -x $X -q $R -osh /dev/shm/o_$mode.sam 2> gpurun_out/r03_cli10k_$mode.txt
  echo "== $mode"; grep stats gpurun_out/r03_cli10k_$mode.txt | cut -c1-230; grep ticket gpurun_out/r03_cli10k_$mode.txt | sed -n '8,10p' | cut -c1-160
done
cmp <(grep -v "^@PG" /dev/shm/o_dev.sam) <(grep -v "^@PG" /dev/shm/o_host.sam) && echo identical
rm -f /dev/shm/o_*.sam
