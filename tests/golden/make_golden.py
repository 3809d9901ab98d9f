#!/usr/bin/env python3
"""Regenerates tests/golden/ from the REAL reference (oracle/_ref/yaha, built from /root/reference by
oracle/Makefile).  Inputs come from this repo's own seeded simulator (tools/yaha_sim.cpp); outputs are what the
reference binary printed.  Only data is stored here -- no reference source in any form.

    python tests/golden/make_golden.py        # needs /root/reference (development container only)
"""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SIM = os.path.join(ROOT, "tools", "yaha_sim")
REF = os.path.join(ROOT, "oracle", "_ref", "yaha")

READ_SETS = {  # name -> simulator arguments
    "r1k.fa": ["--seed", "5", "--n", "120", "--len", "1000", "--div", "0.017", "--edges", "--withN", "0.03"],
    "rchim.fa": ["--seed", "22", "--n", "300", "--len", "500", "--div", "0.02", "--chimeric", "0.4", "--withN", "0.05"],
    "rq.fq": ["--seed", "23", "--n", "250", "--len", "200", "--div", "0.034", "--fastq", "--len-jitter", "150", "--chimeric", "0.1"],
    "r100.fa": ["--seed", "24", "--n", "400", "--len", "100", "--div", "0.007", "--len-jitter", "88"],
    "r10k.fa": ["--seed", "21", "--n", "12", "--len", "10000", "--div", "0.034"],
}
RUNS = [  # (golden name, read set, output flag, extra reference options)
    ("r1k_default", "r1k.fa", "-osh", []),
    ("r1k_X10_MD20", "r1k.fa", "-osh", ["-X", "10", "-MD", "20"]),
    ("r1k_H20", "r1k.fa", "-osh", ["-H", "20"]),
    ("r1k_AGSN", "r1k.fa", "-osh", ["-AGS", "N"]),
    ("r1k_BW8_G80", "r1k.fa", "-osh", ["-BW", "8", "-G", "80", "-MNO", "10"]),
    ("r1k_o8", "r1k.fa", "-o8", []),
    ("rchim_default", "rchim.fa", "-osh", []),
    ("rchim_FBS", "rchim.fa", "-oss", ["-FBS", "Y"]),
    ("rchim_OQCN", "rchim.fa", "-osh", ["-OQC", "N"]),
    ("rchim_M15_P08", "rchim.fa", "-osh", ["-M", "15", "-P", "0.8"]),
    ("rchim_BW3_G20", "rchim.fa", "-osh", ["-BW", "3", "-G", "20"]),
    ("rq_default", "rq.fq", "-osh", []),
    ("rq_scores", "rq.fq", "-osh", ["-GOC", "3", "-GEC", "1", "-RC", "2", "-MS", "2", "-FBS", "Y", "-PRL", "0.5", "-PSS", "0.5"]),
    ("r100_default", "r100.fa", "-osh", []),
    ("r10k_default", "r10k.fa", "-osh", []),
]


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def gz(src, dst):
    with open(src, "rb") as f, gzip.GzipFile(dst, "wb", mtime=0) as g:
        shutil.copyfileobj(f, g)


def main():
    assert os.path.exists(REF), "build the reference first: make -C oracle ref"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", SIM, os.path.join(ROOT, "tools", "yaha_sim.cpp")])
    meta = {"reference": "GregoryFaust/yaha v0.1.83 (oracle/_ref/yaha)", "index": {}, "runs": {}}
    with tempfile.TemporaryDirectory() as td:
        g = os.path.join(td, "genome_small.fa")
        subprocess.check_call([SIM, "genome", "--seed", "11", "--out", g, "--seqs", "3", "--len", "300000", "--repeat-frac", "0.35"])
        gz(g, os.path.join(HERE, "genome_small.fa.gz"))
        for L, H in ((11, None), (8, "20")):
            args = [REF, "-g", g, "-L", str(L)] + (["-H", H] if H else [])
            subprocess.check_call(args, stderr=subprocess.DEVNULL)
            x = [f for f in os.listdir(td) if f.startswith("genome_small.X%02d" % L)][0]
            meta["index"][x] = {"sha256": sha(os.path.join(td, x)), "size": os.path.getsize(os.path.join(td, x))}
        meta["index"]["genome_small.nib2"] = {"sha256": sha(os.path.join(td, "genome_small.nib2")), "size": os.path.getsize(os.path.join(td, "genome_small.nib2"))}
        idx = os.path.join(td, "genome_small.X11_01_65525S")
        for name, sargs in READ_SETS.items():
            p = os.path.join(td, name)
            subprocess.check_call([SIM, "reads", "--genome", g, "--out", p] + sargs)
            gz(p, os.path.join(HERE, name + ".gz"))
        for gname, rs, oflag, extra in RUNS:
            out = os.path.join(td, gname + ".out")
            subprocess.check_call([REF, "-x", idx, "-q", os.path.join(td, rs), oflag, out] + extra, stderr=subprocess.DEVNULL)
            lines = [l for l in open(out).read().split("\n") if not l.startswith("@PG")]
            with gzip.GzipFile(os.path.join(HERE, gname + ".out.gz"), "wb", mtime=0) as gzf:
                gzf.write("\n".join(lines).encode())
            meta["runs"][gname] = {"reads": rs, "oflag": oflag, "extra": extra, "lines": len(lines)}
    json.dump(meta, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    sys.exit(main())
