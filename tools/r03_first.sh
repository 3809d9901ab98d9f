#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
C=/tmp/yaha_bench_cache; X=$C/g3100m_s42.X15_01_65525S; R10=$C/cli10k.fa; R1=$C/e2e_n1048576_l1000_s3000.fa
[ -f $R10 ] || tools/yaha_sim reads --genome $C/g3100m_s42.fa --out $R10 --seed 77 --n 32768 --len 10000 --div 0.034
[ -f $R1 ] || tools/yaha_sim reads --genome $C/g3100m_s42.fa --out $R1 --seed 3000 --n 1048576 --len 1000 --div 0.017
yaha_amd/csrc/yaha -x $X -q $R10 -osh /dev/shm/o.sam 2>/dev/null
for rep in 1 2; do for m in 0 1 2; do for R in $R10 $R1; do
  sleep 25
  YAHA_FIRST=$m YAHA_STATS=1 yaha_amd/csrc/yaha -x $X -q $R -osh /dev/shm/o.sam 2>&1 | grep stats | sed "s/^/first=$m $(basename $R | cut -c1-8): /" | cut -c1-215
done; done; done
rm -f /dev/shm/o.sam
