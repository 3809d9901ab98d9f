"""End-to-end timing of the yaha command line on the bench genome (run on the GPU box after bench.py has built the cache)."""
import subprocess, time, sys, os
root = os.environ.get("GRAFT_REPO_ROOT", ".")
X = "/tmp/yaha_bench_cache/g100m_s42.X15_01_65525S"; G = "/tmp/yaha_bench_cache/g100m_s42.fa"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
R = "/tmp/yaha_bench_cache/cli_reads_%d.fa" % N
if not os.path.exists(R):
    subprocess.check_call([os.path.join(root, "tools/yaha_sim"), "reads", "--genome", G, "--out", R, "--seed", "77", "--n", str(N), "--len", "1000", "--div", "0.017"])
outs = []
tiny = "/tmp/yaha_bench_cache/cli_tiny.fa"
with open(R) as f, open(tiny, "w") as g:
    for _ in range(32): g.write(f.readline())
for extra in (["-t", "16", "-ctx", "2"], ["-t", "16", "-ctx", "1"], ["-t", "16", "-ctx", "2", "-batch", "8192"], ["-t", "16", "-ctx", "2", "-batch", "16384"], ["-ctx", "2"]):
    # a process that follows one with a large device footprint waits seconds for the driver to scrub the freed memory: let a tiny run absorb that
    subprocess.run([os.path.join(root, "yaha_amd/csrc/yaha"), "-x", X, "-q", tiny, "-osh", "/tmp/tiny.sam"], stderr=subprocess.DEVNULL, check=True)
    o = "/tmp/out_%d.sam" % len(outs); s = time.time()
    subprocess.run([os.path.join(root, "yaha_amd/csrc/yaha"), "-x", X, "-q", R, "-osh", o] + extra, stderr=subprocess.DEVNULL, check=True)
    dt = time.time() - s; print(" ".join(extra), ": wall %.2f s -> %.0f reads/s end to end (process start, index mmap + upload included)" % (dt, N / dt)); outs.append(o)
a = [l for l in open(outs[0]) if not l.startswith("@PG")]
for o in outs[1:]:
    print("identical minus @PG:", a == [l for l in open(o) if not l.startswith("@PG")])
