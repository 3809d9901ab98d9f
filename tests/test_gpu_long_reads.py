"""GPU tier, the BASELINE configs that need long reads on a human-like genome at the reference's default seed length (-L 15):

  * config 5 -- CGR-like contigs of ~30 kbp (2..6 rearranged segments: deletions, tandem duplications, inversions, distal pieces) at 0 / 1 / 4 %
    divergence, run with -OQC Y -FBS Y (testdata/README.txt:25-29), plus one read of exactly 32 000 bases (the longest the reference takes,
    AlignArgs.c:82) and one of 32 001 (skipped with a warning, Query.c:148-156);
  * config 5 -- the 500-mer split-read sets (testdata/README.txt:9-23): reads sampled at 2 % from SV event contigs -- deletions, tandem duplications, inversions and
    distal insertions of 100 .. 10 000 bp (RandomSV_Events.sim:1-4) and insertions of a 300-bp repeat-family copy flanked on both sides (Alu_Insertions.sim) --
    run with -OQC Y -FBS Y: the break point penalties and the -M 25 / -MNO corner of GraphPath.cpp:897-1086 (three segments inside 500 bases);
  * config 3 -- 10 kbp reads at the realised divergence of the bundled "E10" sets (3.4 %).

The whole `yaha` command line of this repo (HIP hot path) against the REAL reference binary when it travelled with the snapshot (oracle/_ref/yaha),
else against the oracle port.  The reference writes the reads of a multi-threaded run in completion order: records are compared per
read, in their order within the read; this repo's output must also be in input order.
"""
import os
import subprocess

import pytest

import oracle
import yaha_amd as ya
from conftest import ROOT, strip_pg

pytestmark = pytest.mark.gpu
SIM = os.path.join(ROOT, "tools", "yaha_sim")


@pytest.fixture(scope="module")
def g40(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("g40"))
    g = os.path.join(d, "g.fa")
    subprocess.check_call([SIM, "genome", "--seed", "1234", "--out", g, "--seqs", "12", "--len", "40000000", "--repeat-frac", "0.45", "--nrun", "3", "--lowcomplex", "6", "--repeat-bed", os.path.join(d, "repeats.bed")])
    ya.build_index(["-g", g, "-L", "15"])
    return d, g, os.path.join(d, "g.X15_01_65525S")


def by_read(lines):
    """header lines, {QNAME: [records in order]}, QNAMEs in order of first appearance"""
    head, recs, order = [], {}, []
    for l in lines:
        if not l:
            continue
        if l.startswith("@"):
            head.append(l)
            continue
        q = l.split("\t", 1)[0]
        if q not in recs:
            recs[q] = []
            order.append(q)
        recs[q].append(l)
    return head, recs, order


def referee(index, reads, extra, out):
    """SAM of the real reference (all host cores) or, without the binary, of the oracle port through the product's host stages."""
    if oracle.have_reference():
        oracle.run_reference(["-x", index, "-q", reads, "-osh", out, "-t", str(min(64, os.cpu_count() or 1))] + extra)
        return strip_pg(open(out, newline="").read())
    text = []
    with ya.Session(["-x", index, "-q", reads, "-osh", "stdout"] + extra) as s:
        text.append(s.header())
        while True:
            b = s.next_batch(64)
            if b.n_reads == 0:
                break
            r, _own = oracle.run(s.index, s.params, b, threads=min(64, os.cpu_count() or 1))
            text.append(s.emit(r))
    return strip_pg("".join(text))


def names_in(path):
    return [l[1:].split()[0].rstrip("\n") for l in open(path) if l.startswith(">")]


def check(index, reads, extra, tmp_path, min_records, skipped=()):
    mine_out, ref_out = str(tmp_path / "mine.sam"), str(tmp_path / "ref.sam")
    subprocess.check_call([ya.CLI_PATH, "-x", index, "-q", reads, "-osh", mine_out, "-t", "8", "-batch", "32"] + extra, stderr=subprocess.DEVNULL)
    mh, mr, morder = by_read(strip_pg(open(mine_out, newline="").read()))
    rh, rr, _ = by_read(referee(index, reads, extra, ref_out))
    assert mh == rh
    assert sum(len(v) for v in rr.values()) >= min_records
    assert set(mr) == set(rr), "different sets of aligned reads"
    bad = [q for q in rr if rr[q] != mr[q]]
    assert not bad, "records differ for %d reads, e.g. %s" % (len(bad), bad[0])
    pos = {q: i for i, q in enumerate(names_in(reads))}
    assert [pos[q] for q in morder] == sorted(pos[q] for q in morder), "output is not in input order"
    for q in skipped:
        assert q not in mr and q not in rr
    return mr


@pytest.mark.parametrize("div", ["0", "0.01", "0.04"])
def test_cgr_like_30kbp_contigs_oqc_fbs(g40, tmp_path, div):
    d, g, index = g40
    reads = str(tmp_path / "cgr.fa")
    subprocess.check_call([SIM, "reads", "--genome", g, "--out", reads, "--seed", "77", "--n", "40", "--len", "30000", "--div", div, "--cgr", "5"])
    skipped = ()
    if div == "0.01":
        # the length limits: exactly 32 000 bases is aligned, 32 001 is skipped
        seq = "".join(l.strip() for l in open(g).read(3000000).split(">")[1].split("\n")[1:])
        a = next(k for k in range(100000, 2000000, 50000) if "N" not in seq[k:k + 32000])
        b = next(k for k in range(a + 50000, 2500000, 50000) if "N" not in seq[k:k + 32001])
        with open(reads, "a") as f:
            f.write(">len32000\n%s\n>len32001\n%s\n" % (seq[a:a + 32000], seq[b:b + 32001]))
        skipped = ("len32001",)
    mr = check(index, reads, ["-OQC", "Y", "-FBS", "Y"], tmp_path, min_records=80, skipped=skipped)
    if div == "0.01":
        assert "len32000" in mr and "\t32000M\t" in mr["len32000"][-1]
    # split contigs: most reads come back as several primary pieces (YP > 1)
    multi = sum(1 for v in mr.values() if any("YP:i:1" not in l for l in v))
    assert multi >= 20


def test_reads_from_inside_repeat_families_reach_the_filters_last_class(g40, tmp_path):
    """Reads drawn straight from the genome around copies of its Alu-like families (16 families of ~900 copies at 4-13 % divergence): a read that holds one or two
    copies chains with hundreds of the others -- 450 to 1 000 clumps -- which is the device post-filter's LAST class (449 .. 1 792 clumps: the sort on the wave in 64 KB
    of LDS, oqc_stage.h).  Until round 6 that class was only ever compared with the host's compilation of the same routine; here the reads go through the command line
    (-OQC Y -FBS Y, the filter on the device) against the reference itself, after the API has shown that such reads are in the set and that none was handed back to
    the host unfiltered."""
    import random
    d, g, index = g40
    seqs, name = {}, None
    for l in open(g):
        if l.startswith(">"):
            name = l[1:].split()[0]; seqs[name] = []
        else:
            seqs[name].append(l.strip())
    seqs = {k: "".join(v) for k, v in seqs.items()}
    bed = [l.split("\t") for l in open(os.path.join(d, "repeats.bed")).read().split("\n") if l]
    rnd = random.Random(6)
    reads = str(tmp_path / "inside_repeats.fa"); n_written = 0
    with open(reads, "w") as f:
        for i, (c, s0, e0, _st, fam, _dv) in enumerate(rnd.sample(bed, 260)):
            mid = (int(s0) + int(e0)) // 2; a = max(0, mid - rnd.randrange(300, 700)); r = seqs[c][a:a + 1000]
            if len(r) == 1000 and "N" not in r:
                f.write(">w%d_%s\n%s\n" % (i, fam, r)); n_written += 1
    assert n_written >= 200
    with ya.Session(["-x", index, "-q", reads, "-FBS", "Y"]) as s:
        b = s.next_batch(4096)
        with ya.Context(s.index, s.params) as c:
            c.upload(b); c.run(); r = c.collect()
            n = [r.clump_start[i + 1] - r.clump_start[i] for i in range(r.n_reads)]
            c.set_postfilter(s); f = c.postfilter()
            unfiltered = sum(1 for k in range(int(f.n_clumps)) if f.clumps[k].primaryCount == 0xFFFF)
    big = [x for x in n if x >= 449]
    assert len(big) >= 5 and max(n) <= 1792, "clumps a read: %s" % sorted(n)[-12:]
    assert unfiltered == 0                                                       # every read, the largest included, was filtered on the device
    check(index, reads, ["-OQC", "Y", "-FBS", "Y"], tmp_path, min_records=n_written)


def test_10kbp_reads_at_default_seed_length(g40, tmp_path):
    d, g, index = g40
    reads = str(tmp_path / "r10k.fa")
    subprocess.check_call([SIM, "reads", "--genome", g, "--out", reads, "--seed", "78", "--n", "192", "--len", "10000", "--div", "0.034"])
    check(index, reads, [], tmp_path, min_records=192)


def sv_events(d, n_ins=40):
    """An events file in the reference's two line formats: RandomSV_Events.sim:1-4 as it stands, plus INS lines (Alu_Insertions.sim) for the least diverged full-length
    Alu-like copies of the genome."""
    rows = [l.split("\t") for l in open(os.path.join(d, "repeats.bed")).read().split("\n") if l]
    rows = sorted((r for r in rows if int(r[2]) - int(r[1]) >= 295), key=lambda r: (float(r[5]), r[0], int(r[1])))[:n_ins]
    path = os.path.join(d, "events.sim")
    with open(path, "w") as f:
        f.write("DEL\t100\t10000\t100\nDUP\t100\t10000\t100\nINR\t100\t10000\t100\nINV\t100\t10000\t100\n")
        for r in rows:
            f.write("INS\t%s\t%s\t%s\t%s\t%s\n" % (r[0], r[1], r[2], r[3], r[4]))
    return path


@pytest.mark.parametrize("opts", [["-OQC", "Y", "-FBS", "Y"], []])
def test_sv_500mers_oqc_fbs(g40, tmp_path, opts):
    """BASELINE config 5's split-read sets: every size of RandomSV_Events.sim (100 .. 10 000 in steps of 100, four event types) once, 40 repeat-family insertions
    once each per size step, 500-mers at 2 %, coverage 1.5 -- about 7 400 reads, a sixth of them across a break point (the others lie inside the larger events'
    contigs, as in the reference's sets)."""
    d, g, index = g40
    reads = str(tmp_path / "sv.fa")
    subprocess.check_call([SIM, "sv", "--genome", g, "--events", sv_events(d), "--out", reads, "--seed", "91", "--per", "1", "--cov", "1.5", "--div", "0.02"], stderr=subprocess.DEVNULL)
    n = len(names_in(reads))
    assert n >= 5000
    mine_out, ref_out = str(tmp_path / "mine.sam"), str(tmp_path / "ref.sam")
    subprocess.check_call([ya.CLI_PATH, "-x", index, "-q", reads, "-osh", mine_out] + opts, stderr=subprocess.DEVNULL)
    mh, mr, morder = by_read(strip_pg(open(mine_out, newline="").read()))
    rh, rr, _ = by_read(referee(index, reads, opts, ref_out))
    assert mh == rh and set(mr) == set(rr)
    bad = [q for q in rr if rr[q] != mr[q]]
    assert not bad, "records differ for %d reads, e.g. %s" % (len(bad), bad[0])
    pos = {q: i for i, q in enumerate(names_in(reads))}
    assert [pos[q] for q in morder] == sorted(pos[q] for q in morder), "output is not in input order"
    # the sets are about split reads: a good share of the reads comes back in more than one primary piece, and the three-segment reads (flank, repeat copy, flank) exist
    multi = sum(1 for v in mr.values() if any("YP:i:1" not in l for l in v))
    assert multi >= n // 10
    assert any(q.startswith("sv_AluLike") and sum(1 for l in v if "YS:i:" in l) >= 2 for q, v in mr.items())
