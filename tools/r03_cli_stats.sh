#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
C=/tmp/yaha_bench_cache; X=$C/g3100m_s42.X15_01_65525S; R=$C/e2e_n1048576_l1000_s3000.fa
[ -f $R ] || tools/yaha_sim reads --genome $C/g3100m_s42.fa --out $R --seed 3000 --n 1048576 --len 1000 --div 0.017
yaha_amd/csrc/yaha -x $X -q $R -osh /dev/shm/o.sam 2>/dev/null; sleep 20
YGPU_STATS=1 YAHA_TIMING=1 yaha_amd/csrc/yaha -x $X -q $R -osh /dev/shm/o.sam 2> gpurun_out/r03_cli_stats.txt
grep -c "run:" gpurun_out/r03_cli_stats.txt; grep "run:" gpurun_out/r03_cli_stats.txt | awk '{for(i=1;i<=NF;i++) if ($i=="attempts") a[$(i+1)]++; if ($0 ~ /ranges/) {match($0,/ranges [0-9]+/); r[substr($0,RSTART,RLENGTH)]++}} END {for (k in a) print "attempts", k, a[k]; for (k in r) print k, r[k]}'
grep "run:" gpurun_out/r03_cli_stats.txt | awk '{match($0,/rc 0, [0-9.]+ ms/); print substr($0,RSTART+6,RLENGTH-9)}' | sort -n | awk '{a[NR]=$1} END {print "run ms: min", a[1], "median", a[int(NR/2)], "p90", a[int(NR*0.9)], "max", a[NR]}'
grep -E "grows|stats" gpurun_out/r03_cli_stats.txt | tail -12 | cut -c1-200
rm -f /dev/shm/o.sam
