"""One-off validation at the bench workload's scale: the whole `yaha` command line of this repo against the real reference binary
(oracle/_ref/yaha) on the bench genome (100 Mbp or, with YAHA_PARITY_GENOME=g3100m_s42, 3.1 Gbp; -L 15) -- SAM identical minus @PG.  Run on the GPU box after bench.py built the cache."""
import os, subprocess, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", ".")
GENOME = os.environ.get("YAHA_PARITY_GENOME", "g100m_s42")            # g3100m_s42 = G-hg18scale (bench.py's default cache)
X = "/tmp/yaha_bench_cache/%s.X15_01_65525S" % GENOME; G = "/tmp/yaha_bench_cache/%s.fa" % GENOME
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
SETS = (("1kbp", ["--len", "1000", "--div", "0.017", "--chimeric", "0.05"], 1.0), ("10kbp", ["--len", "10000", "--div", "0.034"], 1 / 16), ("150bp", ["--len", "150", "--div", "0.01"], 2.0),
        ("fastq_N_edges_jitter", ["--len", "800", "--div", "0.05", "--fastq", "--withN", "0.3", "--edges", "--len-jitter", "700", "--chimeric", "0.1"], 0.25),
        ("30kbp", ["--len", "30000", "--div", "0.034", "--chimeric", "0.3"], 1 / 128))
print("commit %s, index %s" % (os.environ.get("GIT_HEAD", "?"), GENOME))
SETS = SETS + (("sv500_OQC_FBS", None, 1.0),)                          # BASELINE config 5: SV / repeat-insertion 500-mers (tools/yaha_sim.cpp `sv`), run with -OQC Y -FBS Y
for tag, extra, frac in SETS:
    n = max(16, int(N * frac)); opts = []
    R = "/tmp/yaha_bench_cache/parity_%s_%d.%s" % (tag, n, "fq" if extra and "--fastq" in extra else "fa")
    if extra is None:
        sys.path.insert(0, root); import bench
        full = bench.make_sv_reads("/tmp/yaha_bench_cache", G, GENOME.split("_")[0], 4242, per=3); bench.head_reads(full, R, n); opts = ["-OQC", "Y", "-FBS", "Y"]
        n = sum(1 for l in open(R) if l.startswith(">"))
    else:
        subprocess.check_call([os.path.join(root, "tools/yaha_sim"), "reads", "--genome", G, "--out", R, "--seed", "4242", "--n", str(n)] + extra)
    t = time.time(); subprocess.run([os.path.join(root, "oracle/_ref/yaha"), "-x", X, "-q", R, "-osh", "/tmp/ref.sam", "-t", os.environ.get("REF_THREADS", "32")] + opts, stderr=subprocess.DEVNULL, check=True); tr = time.time() - t
    t = time.time(); subprocess.run([os.path.join(root, "yaha_amd/csrc/yaha"), "-x", X, "-q", R, "-osh", "/tmp/mine.sam"] + opts, stderr=subprocess.DEVNULL, check=True); tm = time.time() - t
    a = [l for l in open("/tmp/ref.sam") if not l.startswith("@PG")]; b = [l for l in open("/tmp/mine.sam") if not l.startswith("@PG")]
    # the reference writes reads in thread-completion order with -t > 1 (Query.c:457-466): compare the header in order, the records as a multiset
    ha, hb = [l for l in a if l.startswith("@")], [l for l in b if l.startswith("@")]
    a, b = sorted(l for l in a if not l.startswith("@")), sorted(l for l in b if not l.startswith("@"))
    print("%s: %d reads, %d SAM records, header identical=%s, records identical (as a multiset)=%s   reference %.1f s (-t REF_THREADS, default 32), this repo %.1f s" % (tag, n, len(a), ha == hb, a == b, tr, tm))
    if a != b:
        print("records only in one of them:", len(set(a) ^ set(b)))
        for k, (u, v) in enumerate(zip(a, b)):
            if u != v: print("first difference at line", k, "\n ref :", u[:300], "\n mine:", v[:300]); break
