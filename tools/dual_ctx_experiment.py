"""Experiment: two device contexts on one GPU, each with its own batch, stepping concurrently (one host thread each)."""
import os, sys, time, threading, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import yaha_amd as ya
X="/tmp/yaha_bench_cache/g100m_s42.X15_01_65525S"; R="/tmp/yaha_bench_cache/g100m_n16384_l1000_s1000.fa"
N=int(sys.argv[1]) if len(sys.argv)>1 else 16384
s = ya.Session(["-x", X, "-q", R])
b = s.next_batch(N)
ctxs=[ya.Context(s.index, s.params) for _ in range(2)]
for c in ctxs: c.upload(b); c.run()
def loop(c, k):
    for _ in range(k): c.run()
K=4
t=time.time(); loop(ctxs[0],K); loop(ctxs[1],K); seq=time.time()-t
t=time.time(); th=[threading.Thread(target=loop,args=(c,K)) for c in ctxs]; [x.start() for x in th]; [x.join() for x in th]; par=time.time()-t
print("%d reads per context: one at a time %.1f ms per batch, two in flight %.1f ms per batch (%.0f -> %.0f reads/s)" % (N, 1e3*seq/(2*K), 1e3*par/(2*K), N*2*K/seq, N*2*K/par))
