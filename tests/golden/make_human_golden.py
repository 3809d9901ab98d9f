#!/usr/bin/env python3
"""Real human sequence in the parity tier (SURVEY.md 8(c) "optional realism fixture", 8(d) "G-pseudo").

hg18 itself is not available, but the reference's testdata holds 1 000 reads of 10 kbp drawn from it
(testdata/hg18L10000E02Q1K.fasta.gz, README.txt:5-9): concatenated they are a 10 Mbp human-like pseudo-reference with
real Alu / L1 / satellite / low-complexity content.  Against it run
  * 2 000 real 1 kbp reads (testdata/hg18L1000E05Q10K) and 2 000 real 200 bp reads (testdata/hg18L200E05Q50K): they come from
    all over hg18, so what they hit in 10 Mbp is repeat families -- the regime of many seed hits and many clumps per read;
  * 1 500 reads of 1 kbp sampled from the pseudo-reference itself by this repo's simulator (they align end to end, some chimeric).
All through the REAL reference binary (oracle/_ref/yaha -L 15, defaults).  Stored: the pseudo-reference, the two real read sets
(data files of the reference's own tests) and the reference's SAM output, gzip-compressed.  The simulated set is regenerated from its
seed by tools/yaha_sim wherever the test runs.

    python tests/golden/make_human_golden.py        # needs /root/reference (development container only)
"""
import gzip
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "human")
SIM = os.path.join(ROOT, "tools", "yaha_sim")
REF = os.path.join(ROOT, "oracle", "_ref", "yaha")
TESTDATA = "/root/reference/testdata"
SIM_ARGS = ["--seed", "404", "--n", "1500", "--len", "1000", "--div", "0.017", "--chimeric", "0.15"]
RUNS = {"hs_real1k": "hs_real1k.fa", "hs_real200": "hs_real200.fa", "hs_sim1k": "hs_sim1k.fa"}


def records(path, limit=None):
    name, seq, n = None, [], 0
    with gzip.open(path, "rt") as f:
        for line in f:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    yield name, "".join(seq)
                    n += 1
                    if limit and n >= limit:
                        return
                name, seq = line[1:], []
            else:
                seq.append(line)
    if name is not None:
        yield name, "".join(seq)


def write_gz(path, text):
    with gzip.GzipFile(path, "wb", compresslevel=9, mtime=0) as g:
        g.write(text.encode())


def main():
    assert os.path.exists(REF), "build the reference first: make -C oracle ref"
    os.makedirs(OUT, exist_ok=True)
    # pseudo-reference: ten sequences of 100 consecutive 10 kbp reads each
    reads = [s for _, s in records(os.path.join(TESTDATA, "hg18L10000E02Q1K.fasta.gz"))]
    assert len(reads) == 1000
    fa = []
    for k in range(10):
        seq = "".join(reads[100 * k:100 * (k + 1)])
        fa.append(">hs%02d pseudo-reference of 100 hg18 reads of 10 kbp\n" % (k + 1))
        fa.extend(seq[i:i + 100] + "\n" for i in range(0, len(seq), 100))
    genome = "".join(fa)
    write_gz(os.path.join(OUT, "hs_pseudo.fa.gz"), genome)
    for src, dst, n in (("hg18L1000E05Q10K.fasta.gz", "hs_real1k.fa", 2000), ("hg18L200E05Q50K.fasta.gz", "hs_real200.fa", 2000)):
        text = "".join(">%s\n%s\n" % (nm, s) for nm, s in records(os.path.join(TESTDATA, src), n))
        write_gz(os.path.join(OUT, dst + ".gz"), text)
    meta = {"reference": "GregoryFaust/yaha v0.1.83 (oracle/_ref/yaha), -L 15, defaults", "sim_args": SIM_ARGS, "runs": {}}
    with tempfile.TemporaryDirectory() as td:
        g = os.path.join(td, "hs_pseudo.fa")
        open(g, "w").write(genome)
        subprocess.check_call([REF, "-g", g, "-L", "15"], stderr=subprocess.DEVNULL)
        idx = os.path.join(td, "hs_pseudo.X15_01_65525S")
        for f in ("hs_real1k.fa", "hs_real200.fa"):
            with gzip.open(os.path.join(OUT, f + ".gz"), "rb") as gzf:
                open(os.path.join(td, f), "wb").write(gzf.read())
        subprocess.check_call([SIM, "reads", "--genome", g, "--out", os.path.join(td, "hs_sim1k.fa")] + SIM_ARGS)
        for name, rs in RUNS.items():
            out = os.path.join(td, name + ".sam")
            subprocess.check_call([REF, "-x", idx, "-q", os.path.join(td, rs), "-osh", out], stderr=subprocess.DEVNULL)
            lines = [l for l in open(out).read().split("\n") if not l.startswith("@PG")]
            write_gz(os.path.join(OUT, name + ".out.gz"), "\n".join(lines))
            meta["runs"][name] = {"reads": rs, "lines": len(lines), "records": sum(1 for l in lines if l and not l.startswith("@"))}
    json.dump(meta, open(os.path.join(OUT, "human.json"), "w"), indent=1, sort_keys=True)
    print(meta["runs"])
    print({f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT))})


if __name__ == "__main__":
    sys.exit(main())
