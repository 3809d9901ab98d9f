"""Debug aid: the same batch through k_ext_rows (YGPU_ROWS16=0) and k_ext_rows16 (=1); prints the first clumps that differ."""
import os, sys, subprocess, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import yaha_amd as ya

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ln = sys.argv[2] if len(sys.argv) > 2 else "1000"
d = tempfile.mkdtemp()
sim = os.path.join(ROOT, "tools", "yaha_sim")
g = os.path.join(d, "g.fa")
subprocess.check_call([sim, "genome", "--seed", "7", "--out", g, "--seqs", "4", "--len", "2000000", "--repeat-frac", "0.3"])
ya.build_index(["-g", g, "-L", "15"])
reads = os.path.join(d, "r.fa")
subprocess.check_call([sim, "reads", "--genome", g, "--out", reads, "--seed", "5", "--n", str(n), "--len", ln, "--div", "0.03", "--chimeric", "0.05"])
out = {}
with ya.Session(["-x", os.path.join(d, "g.X15_01_65525S"), "-q", reads]) as s:
    b = s.next_batch(n)
    for mode in ("0", "1"):
        os.environ["YGPU_ROWS16"] = mode
        with ya.Context(s.index, s.params) as ctx:
            ctx.upload(b); ctx.run(); out[mode] = ya.result_records(ctx.collect())
a, bb = out["0"], out["1"]
nd = 0
for i, (x, y) in enumerate(zip(a, bb)):
    if x != y:
        nd += 1
        if nd <= 3:
            print("read", i, "clumps", len(x), len(y))
            for k, (cx, cy) in enumerate(zip(x, y)):
                if cx != cy:
                    print("  clump", k)
                    print("   32:", cx[:10], cx[10][:6], "...", cx[10][-6:])
                    print("   16:", cy[:10], cy[10][:6], "...", cy[10][-6:])
                    # first differing op
                    for t, (ox, oy) in enumerate(zip(cx[10], cy[10])):
                        if ox != oy:
                            print("   first op diff at", t, "of", len(cx[10]), len(cy[10]), cx[10][max(0, t - 3):t + 4], cy[10][max(0, t - 3):t + 4]); break
                    break
print("reads", len(a), "differing", nd)
