# what the command line's contexts do at -ctx 4 (YGPU_STATS: attempts of the align stage, ranges, arena sizes, free memory a run)
C=/tmp/yaha_bench_cache; X=$C/g3100m_s42.X15_01_65525S; R1=$C/e2e_n1048576_l1000_s3000.fa
python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
[ -f $R1 ] || tools/yaha_sim reads --genome $C/g3100m_s42.fa --out $R1 --seed 3000 --n 1048576 --len 1000 --div 0.017
for ctx in 3 4; do
  YGPU_STATS=1 YAHA_STATS=1 yaha_amd/csrc/yaha -x $X -q $R1 -osh /dev/shm/o.sam -ctx $ctx 2> /tmp/err_$ctx.txt
  echo "== -ctx $ctx"; grep "stats" /tmp/err_$ctx.txt | cut -c1-600
  grep "run:" /tmp/err_$ctx.txt | awk '{print $0}' | sed -n '1,6p;60,64p' | cut -c1-220
  grep -c "repeated" /tmp/err_$ctx.txt; grep "grows\|left out\|repeated" /tmp/err_$ctx.txt | tail -5 | cut -c1-200
  sleep 20
done
rm -f /dev/shm/o.sam
