import os, subprocess, sys, collections
root = os.environ.get("GRAFT_REPO_ROOT", ".")
X = "/tmp/yaha_bench_cache/g100m_s42.X15_01_65525S"; R = "/tmp/yaha_bench_cache/parity_10kbp_1024.fa"
subprocess.run([os.path.join(root, "oracle/_ref/yaha"), "-x", X, "-q", R, "-osh", "/tmp/ref.sam", "-t", "256"], stderr=subprocess.DEVNULL, check=True)
subprocess.run([os.path.join(root, "yaha_amd/csrc/yaha"), "-x", X, "-q", R, "-osh", "/tmp/mine.sam", "-t", "32"], stderr=subprocess.DEVNULL, check=True)
def recs(p):
    d = collections.defaultdict(list)
    for l in open(p):
        if l.startswith("@"): continue
        f = l.rstrip("\n").split("\t"); d[f[0]].append(f)
    return d
a, b = recs("/tmp/ref.sam"), recs("/tmp/mine.sam")
nd = 0; fieldc = collections.Counter(); cntdiff = 0
for q in a:
    ra, rb = sorted(a[q]), sorted(b.get(q, []))
    if ra == rb: continue
    nd += 1
    if len(ra) != len(rb): cntdiff += 1
    # match by (flag, chr, pos, cigar)
    kb = {(f[1], f[2], f[3], f[5]): f for f in rb}
    for f in ra:
        g = kb.get((f[1], f[2], f[3], f[5]))
        if g is None: fieldc["missing_in_mine"] += 1; continue
        for i in range(len(f)):
            if i < len(g) and f[i] != g[i]: fieldc[(i, f[i].split(":")[0] if i >= 11 else i)] += 1
    if nd <= 3:
        print("read", q, "records ref", len(ra), "mine", len(rb))
        for f in ra: print("  ref ", f[1], f[2], f[3], f[4], f[5][:50], f[11:])
        for f in rb: print("  mine", f[1], f[2], f[3], f[4], f[5][:50], f[11:])
print("reads with differences:", nd, "of", len(a), "; with different record counts:", cntdiff)
print(fieldc.most_common(12))
