"""GPU tier, bounded (< 3 min): the command line against the real reference binary on an index of the bench workload's KIND -- a 400 Mbp repeat-rich genome at
-L 15 (4.3 GB table), built on the device in the test -- so that the paths only a large index reaches are checked by the driver's own test run and not only by
builder-run logs (profiles/*at_scale*): segments of more than 16 384 seed hits (cut by diagonal in k_seg_split, the residue sorted by the library), dead
single-hit fragments counted instead of written, clump slots bounded from the previous batch, several batches in flight on three contexts.  4 096 x 1 kbp
(a tenth chimeric) and 64 x 10 kbp reads; the reference writes in thread-completion order, so records are compared as multisets and this implementation's own
order is checked against the input order."""
import os
import re
import subprocess

import pytest

import oracle
import yaha_amd as ya
from conftest import ROOT, strip_pg

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not oracle.have_reference(), reason="oracle/_ref/yaha did not travel with the snapshot")
def test_bench_scale_index_against_the_reference_binary(tmp_path):
    sim = os.path.join(ROOT, "tools", "yaha_sim")
    d = os.environ.get("YAHA_SCALE_DIR", str(tmp_path))
    g = os.path.join(d, "g400.fa")
    subprocess.check_call([sim, "genome", "--seed", "77", "--out", g, "--seqs", "8", "--len", "400000000", "--repeat-frac", "0.45", "--nrun", "6", "--lowcomplex", "8"])
    ya.build_index(["-g", g, "-L", "15"])                       # .nib2 + index, the index on the device
    idx = os.path.join(d, "g400.X15_01_65525S")
    assert os.path.getsize(idx) > 5 * 10**9
    r1, r2, reads = os.path.join(d, "r1k.fa"), os.path.join(d, "r10k.fa"), os.path.join(d, "reads.fa")
    subprocess.check_call([sim, "reads", "--genome", g, "--out", r1, "--seed", "5", "--n", "4096", "--len", "1000", "--div", "0.017", "--chimeric", "0.1"])
    subprocess.check_call([sim, "reads", "--genome", g, "--out", r2, "--seed", "6", "--n", "64", "--len", "10000", "--div", "0.034"])
    with open(reads, "wb") as o:                                # long reads first: the later 1 kbp batches need more clump slots than twice what the first batch used
        o.write(open(r2, "rb").read()); o.write(open(r1, "rb").read())
    ref_out, out = os.path.join(d, "ref.sam"), os.path.join(d, "mine.sam")
    oracle.run_reference(["-x", idx, "-q", reads, "-osh", ref_out, "-t", "16"])
    p = subprocess.run([ya.CLI_PATH, "-x", idx, "-q", reads, "-osh", out, "-batch", "1024"], env=dict(os.environ, YGPU_TRACE="1"), stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    mine, ref = strip_pg(open(out).read()), strip_pg(open(ref_out).read())
    head = lambda L: [l for l in L if l.startswith("@")]
    recs = lambda L: [l for l in L if l and not l.startswith("@")]
    assert head(mine) == head(ref)
    assert len(recs(mine)) > 4000 and sorted(recs(mine)) == sorted(recs(ref))
    # this implementation writes in input order (= the reference at -t 1)
    ids = [l[1:].split()[0].replace(" ", "_") for l in open(reads) if l.startswith(">")]
    rank = {q: i for i, q in enumerate(ids)}
    order = [rank[l.split("\t")[0]] for l in recs(mine)]
    assert order == sorted(order)
    # the large-index paths were really taken: segments too long for one workgroup sort were cut into buckets
    err = p.stderr.decode()
    m = re.findall(r"hit sort: (\d+) long segments cut into buckets", err)
    assert m and max(int(x) for x in m) > 0, "no segment of more than 16 384 hits: the genome is too small for what this test is for"
    for f in (g, idx, os.path.join(d, "g400.nib2"), reads, r1, r2):
        os.remove(f)
