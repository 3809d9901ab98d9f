#!/bin/bash
# the command line on 1 M reads: reads with more than N clumps handed to the host's filter (default 448), runs 25 s apart
cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
C=/tmp/yaha_bench_cache; X=$C/g3100m_s42.X15_01_65525S; R=$C/e2e_n1048576_l1000_s3000.fa
[ -f $R ] || tools/yaha_sim reads --genome $C/g3100m_s42.fa --out $R --seed 3000 --n 1048576 --len 1000 --div 0.017
yaha_amd/csrc/yaha -x $X -q $R -osh /dev/shm/o.sam 2>/dev/null
for m in 448 224 112 448 224; do
  sleep 25
  YGPU_OQC_MAX=$m YAHA_STATS=1 yaha_amd/csrc/yaha -x $X -q $R -osh /dev/shm/o.sam 2>&1 | grep "stats" | sed "s/^/[max $m] /" | cut -c1-230
done
rm -f /dev/shm/o.sam
