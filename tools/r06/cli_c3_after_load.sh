# the 10 kbp command line right behind two minutes of the hot path at full load, and again after a rest: is the bench's slow last leg the device's state (clocks, memory being scrubbed)?
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
C=/tmp/yaha_bench_cache; X=$C/g3100m_s42.X15_01_65525S
Q=$C/e2e_n32768_l10000_s3100.fa; [ -f $Q ] || tools/yaha_sim reads --genome $C/g3100m_s42.fa --out $Q --seed 3100 --n 32768 --len 10000 --div 0.034
st() { grep -o "total_ms[^,]*, \"steady_reads_per_s\": [0-9]*\|\"run\": [0-9.]*\|filter_thread_ms_per_batch\": [0-9.]*" /tmp/err.txt | tr '\n' ' '; echo; }
yaha_amd/csrc/yaha -x $X -q $Q -osh /dev/shm/o.sam 2>/dev/null
sleep 20; YAHA_STATS=1 yaha_amd/csrc/yaha -x $X -q $Q -osh /dev/shm/o.sam 2> /tmp/err.txt; echo "rested: $(st)"
python bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 5 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('load:', round(j['value']), round(j['ms_per_step'],2))"
YAHA_STATS=1 yaha_amd/csrc/yaha -x $X -q $Q -osh /dev/shm/o.sam 2> /tmp/err.txt; echo "right behind the load: $(st)"
sleep 5; YAHA_STATS=1 yaha_amd/csrc/yaha -x $X -q $Q -osh /dev/shm/o.sam 2> /tmp/err.txt; echo "5 s later: $(st)"
sleep 20; YAHA_STATS=1 yaha_amd/csrc/yaha -x $X -q $Q -osh /dev/shm/o.sam 2> /tmp/err.txt; echo "20 s later: $(st)"
rm -f /dev/shm/o.sam
