"""Validation against the real reference on the bench genome for several option sets (records compared as multisets, see big_parity.py)."""
import os, subprocess, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", ".")
GENOME = os.environ.get("YAHA_PARITY_GENOME", "g100m_s42")
X = "/tmp/yaha_bench_cache/%s.X15_01_65525S" % GENOME; G = "/tmp/yaha_bench_cache/%s.fa" % GENOME
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
R = "/tmp/yaha_bench_cache/parity_opts_%d.fa" % N
subprocess.check_call([os.path.join(root, "tools/yaha_sim"), "reads", "--genome", G, "--out", R, "--seed", "99", "--n", str(N), "--len", "1000", "--div", "0.03", "--chimeric", "0.2"])
sets = [[], ["-OQC", "N"], ["-FBS", "Y"], ["-AGS", "N"], ["-X", "10", "-MD", "20"], ["-BW", "8", "-G", "80"], ["-BW", "3", "-G", "20"], ["-M", "15", "-P", "0.8"], ["-H", "200"],
        ["-G", "15"], ["-M", "15"], ["-M", "16", "-G", "3"], ["-M", "30", "-G", "120"], ["-GOC", "9", "-GEC", "3", "-RC", "1"], ["-MS", "2", "-X", "40"], ["-oss"]]
ok = True
print("commit %s, index %s" % (os.environ.get("GIT_HEAD", "?"), GENOME))
for extra in sets:
    oflag = "-oss" if extra == ["-oss"] else "-osh"; ex = [] if extra == ["-oss"] else extra
    subprocess.run([os.path.join(root, "oracle/_ref/yaha"), "-x", X, "-q", R, oflag, "/tmp/ref.sam", "-t", os.environ.get("REF_THREADS", "32")] + ex, stderr=subprocess.DEVNULL, check=True)
    t = time.time(); subprocess.run([os.path.join(root, "yaha_amd/csrc/yaha"), "-x", X, "-q", R, oflag, "/tmp/mine.sam"] + ex, stderr=subprocess.DEVNULL, check=True); tm = time.time() - t
    a = sorted(l for l in open("/tmp/ref.sam") if not l.startswith("@PG")); b = sorted(l for l in open("/tmp/mine.sam") if not l.startswith("@PG"))
    print("%-28s records %6d identical=%s  (%.1f s)" % (" ".join(extra) or "(defaults)", len(a), a == b, tm)); ok &= a == b
print("ALL IDENTICAL" if ok else "DIFFERENCES FOUND")
