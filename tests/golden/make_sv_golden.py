#!/usr/bin/env python3
"""Adds BASELINE config 5's split-read fixture to tests/golden/: 500-mers sampled from SV event contigs (deletion, tandem duplication, inversion, distal
insertion, and a 300-bp repeat-family insertion flanked on both sides) as the reference's testdata/README.txt:9-23 describes its RandomSV_Events / Alu_Insertions
sets, made by this repo's simulator (tools/yaha_sim.cpp, `sv` mode) on genome_small, and what the REAL reference (oracle/_ref/yaha) printed for them with
-OQC Y -FBS Y and with the defaults.  Only data is stored -- inputs and the reference's outputs.

    python tests/golden/make_sv_golden.py        # needs /root/reference (development container only)
"""
import gzip
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

EVENTS = "DEL\t100\t2100\t400\nDUP\t100\t2100\t400\nINR\t100\t2100\t400\nINV\t100\t2100\t400\n"      # the line format of testdata/RandomSV_Events.sim:1-4
SV_ARGS = ["--seed", "31", "--per", "2", "--pad", "500", "--len", "500", "--cov", "3", "--div", "0.02"]
RUNS = [("rsv_OQC_FBS", "rsv.fa", "-osh", ["-OQC", "Y", "-FBS", "Y"]), ("rsv_default", "rsv.fa", "-osh", [])]


def gz(src, dst):
    with open(src, "rb") as f, gzip.GzipFile(dst, "wb", mtime=0) as g:
        shutil.copyfileobj(f, g)


def main():
    assert os.path.exists(REF), "build the reference first: make -C oracle ref"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", SIM, os.path.join(ROOT, "tools", "yaha_sim.cpp")])
    meta = json.load(open(os.path.join(HERE, "golden.json")))
    with tempfile.TemporaryDirectory() as td:
        g = os.path.join(td, "genome_small.fa")
        bed = os.path.join(td, "repeats.bed")
        subprocess.check_call([SIM, "genome", "--seed", "11", "--out", g, "--seqs", "3", "--len", "300000", "--repeat-frac", "0.35", "--repeat-bed", bed])
        with gzip.open(os.path.join(HERE, "genome_small.fa.gz"), "rb") as f:
            assert f.read() == open(g, "rb").read(), "the simulator no longer reproduces genome_small"
        # INS lines (the format of testdata/Alu_Insertions.sim): the six least diverged, full-length Alu-like copies the genome holds
        rows = [l.split("\t") for l in open(bed).read().split("\n") if l]
        rows = sorted((r for r in rows if int(r[2]) - int(r[1]) >= 295), key=lambda r: (float(r[5]), r[0], int(r[1])))[:6]
        sim = EVENTS + "".join("INS\t%s\t%s\t%s\t%s\t%s\n" % (r[0], r[1], r[2], r[3], r[4]) for r in rows)
        open(os.path.join(HERE, "rsv_events.sim"), "w").write(sim)
        reads = os.path.join(td, "rsv.fa")
        subprocess.check_call([SIM, "sv", "--genome", g, "--events", os.path.join(HERE, "rsv_events.sim"), "--out", reads] + SV_ARGS)
        gz(reads, os.path.join(HERE, "rsv.fa.gz"))
        subprocess.check_call([REF, "-g", g, "-L", "11"], stderr=subprocess.DEVNULL)
        idx = os.path.join(td, "genome_small.X11_01_65525S")
        for gname, rs, oflag, extra in RUNS:
            out = os.path.join(td, gname + ".out")
            subprocess.check_call([REF, "-x", idx, "-q", os.path.join(td, rs), oflag, out] + extra, stderr=subprocess.DEVNULL)
            lines = [l for l in open(out).read().split("\n") if not l.startswith("@PG")]
            with gzip.GzipFile(os.path.join(HERE, gname + ".out.gz"), "wb", mtime=0) as gzf:
                gzf.write("\n".join(lines).encode())
            meta["runs"][gname] = {"reads": rs, "oflag": oflag, "extra": extra, "lines": len(lines)}
            split = sum(1 for l in lines if l and not l.startswith("@") and "YP:i:1" not in l.split("\t", 11)[-1])
            print(gname, len(lines), "lines;", split, "records of reads printed in more than one piece")
    meta["sv_args"] = SV_ARGS
    json.dump(meta, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    sys.exit(main())
