"""Runs the host stages (argument handling, .nib2 / index builder and loader, FASTA/FASTQ reader incl. the block-streaming source, OQC/FBS filter, SAM / Blast8
text, C-ABI session calls) of the sanitizer build (libyaha_host_asan.so: -fsanitize=address,undefined) over the golden inputs, the oracle supplying the
hot-path results.  Started by tests/test_host_sanitizers.py with LD_PRELOAD=libasan/libubsan and YAHA_HIP_LIB pointing at the sanitizer build; any
sanitizer report aborts the process."""
import gzip, hashlib, json, os, shutil, sys, tempfile, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import yaha_amd as ya
import oracle
GOLD = os.path.join(ROOT, "tests", "golden")
assert "asan" in ya.LIB_PATH, ya.LIB_PATH
meta = json.load(open(os.path.join(GOLD, "golden.json")))
d = tempfile.mkdtemp()
for f in os.listdir(GOLD):
    if f.endswith(".gz") and not f.endswith(".out.gz") and not f.endswith(".json.gz"):
        with gzip.open(os.path.join(GOLD, f), "rb") as g, open(os.path.join(d, f[:-3]), "wb") as o:
            shutil.copyfileobj(g, o)
sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()
ya.build_index(["-g", os.path.join(d, "genome_small.fa"), "-L", "11"])
ya.build_index(["-g", os.path.join(d, "genome_small.nib2"), "-L", "8", "-H", "20"])
assert sha(os.path.join(d, "genome_small.X11_01_65525S")) == meta["index"]["genome_small.X11_01_65525S"]["sha256"]
assert sha(os.path.join(d, "genome_small.X08_01_00020S")) == meta["index"]["genome_small.X08_01_00020S"]["sha256"]
idx = os.path.join(d, "genome_small.X11_01_65525S")


def run(reads, oflag, extra, batch=97):
    out = []
    with ya.Session(["-x", idx, "-q", reads, oflag, "stdout"] + list(extra)) as s:
        out.append(s.header())
        while True:
            b = s.next_batch(batch)
            if b.n_reads == 0:
                break
            r, _own = oracle.run(s.index, s.params, b, threads=4)
            out.append(s.emit(r))
    return [l for l in "".join(out).split("\n") if not l.startswith("@PG")]


n = 0
for name, r in sorted(meta["runs"].items()):
    with gzip.open(os.path.join(GOLD, name + ".out.gz"), "rb") as g:
        want = g.read().decode().split("\n")
    assert run(os.path.join(d, r["reads"]), r["oflag"], r["extra"] + ["-t", "3"]) == want, name
    n += 1
# the block-streaming source (stdin / pipes) with blocks far smaller than a record
os.environ["YAHA_READ_BLOCK"] = "61"
fifo = os.path.join(d, "in.fifo"); os.mkfifo(fifo)
data = open(os.path.join(d, "rq.fq"), "rb").read()
th = threading.Thread(target=lambda: open(fifo, "wb").write(data)); th.start()
got = run(fifo, "-osh", [], batch=5); th.join()
with gzip.open(os.path.join(GOLD, "rq_default.out.gz"), "rb") as g:
    assert got == g.read().decode().split("\n")
# error paths
for bad in (["-x", "missing.X11_01_65525S", "-q", "none.fa"], ["-x", idx, "-q", os.path.join(d, "r1k.fa"), "-BW", "-3"], ["-q", "x"]):
    try:
        ya.Session(bad); raise SystemExit("bad arguments accepted: %r" % (bad,))
    except RuntimeError:
        pass
with ya.Session(["-x", idx, "-q", os.path.join(d, "r1k.fa")]) as s:
    try:
        ya.Context(s.index, s.params); raise SystemExit("a context without device code?")
    except RuntimeError:
        pass
shutil.rmtree(d)
print("sanitizer run ok: %d golden runs + index builds + streaming reader + error paths" % n)
