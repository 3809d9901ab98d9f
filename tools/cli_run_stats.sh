R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
C=/tmp/yaha_bench_cache; X=$C/g3100m_s42.X15_01_65525S; R1=$C/e2e_n1048576_l1000_s3000.fa
[ -f $R1 ] || tools/yaha_sim reads --genome $C/g3100m_s42.fa --out $R1 --seed 3000 --n 1048576 --len 1000 --div 0.017
yaha_amd/csrc/yaha -x $X -q $R1 -osh /dev/shm/o.sam 2>/dev/null
sleep 20
YGPU_STATS=1 YAHA_STATS=1 yaha_amd/csrc/yaha -x $X -q $R1 -osh /dev/shm/o.sam $CLI_OPTS 2> gpurun_out/r05z_stats.err
grep -c "repeated" gpurun_out/r05z_stats.err; grep "run:" gpurun_out/r05z_stats.err | awk '{print $8, $12, $14}' | sort | uniq -c | sort -rn | head -8
grep "grow\|repeated" gpurun_out/r05z_stats.err | head -20 | cut -c1-220
grep "run:" gpurun_out/r05z_stats.err | sed -n '10,16p' | cut -c1-200
grep "stats" gpurun_out/r05z_stats.err | cut -c1-700
rm -f /dev/shm/o.sam
