#!/bin/bash
# how fast does the 16.7 GB index image reach HBM with 1 / 2 / 4 / 8 uploading threads?  (command line on 65 536 reads, YAHA_TIMING's "contexts up")
cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
C=/tmp/yaha_bench_cache; X=$C/g3100m_s42.X15_01_65525S; R=$C/up.fa; tools/yaha_sim reads --genome $C/g3100m_s42.fa --out $R --seed 5 --n 65536 --len 1000 --div 0.017
yaha_amd/csrc/yaha -x $X -q $R -osh /dev/shm/o.sam 2>/dev/null
for rep in 1 2; do for t in 1 2 4 8; do
  YGPU_UPLOAD_THREADS=$t YAHA_TIMING=1 yaha_amd/csrc/yaha -x $X -q $R -osh /dev/shm/o.sam 2>&1 | grep "contexts up" | sed "s/^/threads $t: /"
done; done
