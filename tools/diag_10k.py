import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import yaha_amd as ya, oracle
X = "/tmp/yaha_bench_cache/g100m_s42.X15_01_65525S"; R = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
with ya.Session(["-x", X, "-q", R]) as s:
    b = s.next_batch(N)
    with ya.Context(s.index, s.params) as ctx:
        ctx.upload(b); ctx.run(); r = ctx.collect()
        got = ya.result_records(r); gc = r.counters.as_dict()
        ro, _own = oracle.run(s.index, s.params, b, threads=64)
        exp = ya.result_records(ro); ec = ro.counters.as_dict()
        bad = [i for i in range(len(got)) if got[i] != exp[i]]
        print("reads", len(got), "differing reads", len(bad))
        for k in gc:
            if gc[k] != ec[k]: print("counter", k, gc[k], ec[k])
        for i in bad[:3]:
            g, e = got[i], exp[i]
            print("read", i, "clumps dev", len(g), "oracle", len(e))
            sg, se = set(g), set(e)
            for c in list(sg - se)[:2]: print("  only device:", c[:10], "nops", len(c[10]))
            for c in list(se - sg)[:2]: print("  only oracle:", c[:10], "nops", len(c[10]))
