# the command line on 32 768 reads of 10 kbp under the sort's rankings: the stats line, the run time a batch, any message of the sort
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
C=/tmp/yaha_bench_cache; X=$C/g3100m_s42.X15_01_65525S; Q=$C/cli10k.fa
[ -f $Q ] || tools/yaha_sim reads --genome $C/g3100m_s42.fa --out $Q --seed 77 --n 32768 --len 10000 --div 0.034
yaha_amd/csrc/yaha -x $X -q $Q -osh /dev/shm/o.sam 2>/dev/null
for v in "" atomic ballots ""; do
  sleep 20
  echo "== YGPU_SORT_RANK=$v"
  YGPU_SORT_RANK=$v YAHA_STATS=1 yaha_amd/csrc/yaha -x $X -q $Q -osh /dev/shm/o.sam 2> /tmp/err.txt; grep -E "hit sort|stats" /tmp/err.txt | cut -c1-600
  md5sum /dev/shm/o.sam
done
rm -f /dev/shm/o.sam
