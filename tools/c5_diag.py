import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, yaha_amd as ya
cache = "/tmp/yaha_bench_cache"
fa, idx = bench.ensure_inputs(cache, 3100, 42)
sv = bench.make_sv_reads(cache, fa, "g3100m", 5000, per=40)
print("reads in set", sum(1 for l in open(sv) if l.startswith(">")), flush=True)
with ya.Session(["-x", idx, "-q", sv]) as s:
    b = s.next_batch(32768)
    with ya.Context(s.index, s.params) as c:
        c.upload(b)
        try:
            c.run(); print("run 1 ok", flush=True); c.run(); print("run 2 ok"); r = c.collect(); print(r.n_clumps, r.n_ops, r.counters.as_dict())
        except Exception as e:
            print("FAILED", e, flush=True)
