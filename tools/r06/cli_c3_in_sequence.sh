# does the 10 kbp command-line leg slow down behind the 1 M-read leg (as inside bench.py), or only under bench.py's process?  Standalone sequence: 1 kbp x 1 M, 20 s, 10 kbp x 32 768 twice
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
C=/tmp/yaha_bench_cache; X=$C/g3100m_s42.X15_01_65525S
R1=$C/e2e_n1048576_l1000_s3000.fa; [ -f $R1 ] || tools/yaha_sim reads --genome $C/g3100m_s42.fa --out $R1 --seed 3000 --n 1048576 --len 1000 --div 0.017
Q=$C/e2e_n32768_l10000_s3100.fa; [ -f $Q ] || tools/yaha_sim reads --genome $C/g3100m_s42.fa --out $Q --seed 3100 --n 32768 --len 10000 --div 0.034
st() { grep -o "total_ms[^,]*, \"steady_reads_per_s\": [0-9]*\|\"run\": [0-9.]*\|filter_thread_ms_per_batch\": [0-9.]*" /tmp/err.txt | tr '\n' ' '; echo; }
YAHA_STATS=1 yaha_amd/csrc/yaha -x $X -q $R1 -osh /dev/shm/o.sam 2> /tmp/err.txt; echo "1 kbp x 1M: $(st)"
sleep 20
for i in 1 2; do YAHA_STATS=1 yaha_amd/csrc/yaha -x $X -q $Q -osh /dev/shm/o.sam 2> /tmp/err.txt; echo "10 kbp run $i: $(st)"; sleep 20; done
# and under a parent that has imported torch and initialised the device (as bench.py has)
python3 - <<PY
import subprocess, os, time, torch
torch.cuda.init(); x = torch.zeros(1, device="cuda")
env = dict(os.environ, YAHA_STATS="1")
for i in range(2):
    t = time.time(); p = subprocess.run(["yaha_amd/csrc/yaha", "-x", "$X", "-q", "$Q", "-osh", "/dev/shm/o.sam"], stderr=subprocess.PIPE, env=env); dt = time.time() - t
    s = [l for l in p.stderr.decode().split("\n") if "stats" in l][0]
    import json; j = json.loads(s[len("[yaha] stats "):]); print("under a torch parent: wall %.2f s, total_ms %s, steady %s, run %s, filter %s" % (dt, j["total_ms"], j["steady_reads_per_s"], j["context_thread_ms_per_batch"]["run"], j["filter_thread_ms_per_batch"]))
    time.sleep(20)
PY
rm -f /dev/shm/o.sam
