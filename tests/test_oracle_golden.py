"""CPU tier: the oracle (oracle/hotpath.cpp) and the product's host stages are pinned against output of the REAL
reference binary (tests/golden/*.out.gz, produced by tests/golden/make_golden.py with oracle/_ref/yaha)."""
import hashlib
import os

import pytest

import oracle
import yaha_amd as ya
from conftest import golden_lines, strip_pg, oflag_args


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def test_nib2_and_index_are_byte_identical_to_the_reference(work, meta):
    assert sha(os.path.join(work, "genome_small.nib2")) == meta["index"]["genome_small.nib2"]["sha256"]
    assert sha(os.path.join(work, "genome_small.X11_01_65525S")) == meta["index"]["genome_small.X11_01_65525S"]["sha256"]


def test_index_sampling_of_overrepresented_kmers_is_byte_identical(work, meta):
    # -L 8 -H 20 forces the Floyd sampling pass with the default-seeded Marsaglia generator (Index.c:271-315)
    ya.build_index(["-g", os.path.join(work, "genome_small.nib2"), "-L", "8", "-H", "20"])
    name = "genome_small.X08_01_00020S"
    assert os.path.getsize(os.path.join(work, name)) == meta["index"][name]["size"]
    assert sha(os.path.join(work, name)) == meta["index"][name]["sha256"]


@pytest.mark.skipif(not oracle.have_reference(), reason="oracle/_ref/yaha not built")
@pytest.mark.parametrize("args", [["-L", "11", "-S", "3"], ["-L", "9", "-S", "9", "-H", "50"], ["-L", "12", "-S", "2"]])
def test_skip_distance_index_matches_the_live_reference(work, tmp_path, args):
    # -S > 1 (Index.c:98-127: steps of S from the sequence start, restart at an absolute multiple of S behind a run of N): the reference binary builds the
    # same genome here and the files must be identical -- the host builder in this tier, the device builder in the GPU tier (test_gpu_parity.py)
    import shutil, subprocess
    a, b = tmp_path / "mine", tmp_path / "ref"; a.mkdir(); b.mkdir()
    for d in (a, b):
        shutil.copy(os.path.join(work, "genome_small.nib2"), str(d / "g.nib2"))
    subprocess.run([ya.CLI_PATH, "-g", str(a / "g.nib2"), "-cpuindex"] + args, stderr=subprocess.DEVNULL, check=True)
    oracle.run_reference(["-g", str(b / "g.nib2")] + args)
    name = [f for f in os.listdir(a) if ".X" in f]
    assert len(name) == 1 and os.path.exists(b / name[0])
    assert sha(str(a / name[0])) == sha(str(b / name[0]))


def run_oracle_pipeline(index, reads, oflag, extra, batch=97, threads=4):
    out = []
    with ya.Session(["-x", index, "-q", reads] + oflag_args(oflag) + list(extra)) as s:
        out.append(s.header())
        while True:
            b = s.next_batch(batch)
            if b.n_reads == 0:
                break
            r, _own = oracle.run(s.index, s.params, b, threads=threads)
            out.append(s.emit(r))
    return strip_pg("".join(out))


def test_every_golden_run_matches(work, index11, meta):
    assert len(meta["runs"]) >= 12
    for name, run in sorted(meta["runs"].items()):
        mine = run_oracle_pipeline(index11, os.path.join(work, run["reads"]), run["oflag"], run["extra"])
        assert mine == golden_lines(name), "oracle + host stages differ from the reference on " + name


def test_batch_size_and_threads_do_not_change_output(work, index11):
    a = run_oracle_pipeline(index11, os.path.join(work, "rchim.fa"), "-osh", [], batch=1000, threads=1)
    b = run_oracle_pipeline(index11, os.path.join(work, "rchim.fa"), "-osh", ["-t", "3"], batch=7, threads=8)
    assert a == b == golden_lines("rchim_default")


def test_nib2_version_1_files_load(work, tmp_path):
    """host/formats.cpp:parseNib2 restates both header layouts of loadBaseSequences (Compress.c:89-128); only version 2 is ever written.  A version-1 file made from
    the golden genome's must give the same sequences, the same header and the same alignments -- and the reference itself must accept the file the test makes."""
    from conftest import nib2_v1_copy
    idx1 = nib2_v1_copy(work, str(tmp_path / "v1"))
    assert run_oracle_pipeline(idx1, os.path.join(work, "rchim.fa"), "-osh", []) == golden_lines("rchim_default")
    if oracle.have_reference():
        ref_out = str(tmp_path / "ref.sam")
        oracle.run_reference(["-x", idx1, "-q", os.path.join(work, "rchim.fa"), "-osh", ref_out])
        assert strip_pg(open(ref_out).read()) == golden_lines("rchim_default")


@pytest.mark.skipif(not oracle.have_reference(), reason="oracle/_ref/yaha not built")
def test_live_reference_on_fresh_reads(work, index11, tmp_path):
    # a read set that is not in the goldens, checked against the reference binary run right now
    import subprocess
    from conftest import ROOT
    reads = str(tmp_path / "fresh.fa")
    subprocess.check_call([os.path.join(ROOT, "tools", "yaha_sim"), "reads", "--genome", os.path.join(work, "genome_small.fa"), "--out", reads,
                           "--seed", "777", "--n", "150", "--len", "700", "--div", "0.05", "--chimeric", "0.2", "--len-jitter", "300"])
    ref_out = str(tmp_path / "ref.sam")
    oracle.run_reference(["-x", index11, "-q", reads, "-osh", ref_out, "-FBS", "Y"])
    assert run_oracle_pipeline(index11, reads, "-osh", ["-FBS", "Y"]) == strip_pg(open(ref_out).read())


def _reads_of(path, n):
    out, name, seq = [], None, []
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            if name is not None:
                out.append((name, "".join(seq)))
                if len(out) == n:
                    return out
            name, seq = line[1:], []
        else:
            seq.append(line)
    out.append((name, "".join(seq)))
    return out[:n]


def _wrap(s, w):
    return "\n".join(s[i:i + w] for i in range(0, len(s), w))


def _quirky_fasta(reads):
    """Input-format corner cases of readNextQuery (Query.c:102-228), on reads that align so that QNAME and SEQ show up in the SAM."""
    t = []
    t.append(">%s with spaces in the id\n%s\n" % (reads[0][0], _wrap(reads[0][1], 60)))          # multi-line, spaces -> '_'
    t.append(">%s\n%s\n" % ("L" * 230, reads[1][1]))                                             # id longer than 200: truncated
    t.append(">%s>gt_inside_id\n%s\n" % (reads[2][0], _wrap(reads[2][1].lower(), 70)))           # '>' in the id line, lower case
    t.append(">tooshort\nACGTAC\n")                                                               # skipped (shorter than wordLen)
    t.append(">%s\n\n\n%s\n\n" % (reads[3][0], _wrap(reads[3][1], 17)))                           # blank lines inside a record
    t.append(">%s\n%s" % (reads[4][0], reads[4][1][:400]) + ">" + "%s_cut\n%s\n" % (reads[4][0], reads[4][1][400:]))   # '>' in mid-line starts a record
    t.append(">%s\r\n%s\r\n" % (reads[5][0], reads[5][1]))                                       # CR is an id character and a base
    t.append(">toolong\n%s\n" % _wrap("ACGT" * 8001, 80))                                          # 32 004 bases: skipped with a warning
    t.append(">%s\n%s\n" % (reads[6][0], reads[6][1]))
    t.append(">%s\n%s" % (reads[7][0], reads[7][1]))                                              # no newline at the end of the input
    return "".join(t)


def _quirky_fastq(reads):
    q = lambda n, c="I": c * n
    t = []
    t.append("@%s\n%s\n+\n%s\n" % (reads[0][0], reads[0][1], "@" + q(len(reads[0][1]) - 1)))      # quality that starts with '@'
    t.append("@%s extra words\n%s\n+%s\n%s\n" % (reads[1][0], _wrap(reads[1][1], 50), reads[1][0], _wrap(q(len(reads[1][1]), "5"), 50)))   # multi-line, id repeated after '+'
    t.append("@mismatch\n%s\n+\n%s\n" % (reads[2][1], q(len(reads[2][1]) - 3)))                   # lengths differ: skipped
    t.append("@%s\n%s\n+\n%s\n" % (reads[3][0], reads[3][1], "".join(chr(33 + (i * 7) % 60) for i in range(len(reads[3][1]))).replace("@", "A")))
    t.append("@noseq\n\n+\n%s\n" % q(12))                                                          # empty sequence but a quality string: skipped, input goes on
    t.append("@%s\n%s\n+\n%s" % (reads[4][0], reads[4][1], q(len(reads[4][1]), "#")))             # no newline at the end
    return "".join(t)


@pytest.mark.skipif(not oracle.have_reference(), reason="oracle/_ref/yaha not built")
@pytest.mark.parametrize("kind", ["fasta", "fastq"])
@pytest.mark.parametrize("source", ["mmap", "pipe_small_blocks"])
def test_reader_quirks_match_the_live_reference(work, index11, tmp_path, kind, source, monkeypatch):
    reads = _reads_of(os.path.join(work, "r1k.fa"), 8)
    path = str(tmp_path / ("quirks." + ("fa" if kind == "fasta" else "fq")))
    with open(path, "w", newline="") as f:
        f.write(_quirky_fasta(reads) if kind == "fasta" else _quirky_fastq(reads))
    ref_out = str(tmp_path / "ref.sam")
    _o, err = oracle.run_reference(["-x", index11, "-q", path, "-osh", ref_out])
    want = strip_pg(open(ref_out, newline="").read())                 # no newline translation: one id ends in a CR
    assert len([l for l in want if l and not l.startswith("@")]) >= (8 if kind == "fasta" else 4)
    if source == "mmap":
        assert run_oracle_pipeline(index11, path, "-osh", [], batch=3) == want
        return
    # the same bytes through a FIFO (the streaming source: stdin, pipes), read in blocks far smaller than a record
    import threading
    monkeypatch.setenv("YAHA_READ_BLOCK", "97")
    fifo = str(tmp_path / "in.fifo")
    os.mkfifo(fifo)
    data = open(path, "rb").read()

    def feed():
        with open(fifo, "wb") as w:
            w.write(data)
    th = threading.Thread(target=feed)
    th.start()
    got = run_oracle_pipeline(index11, fifo, "-osh", [], batch=2)
    th.join()
    # the @PG-free output is identical; the reference's file name is in @PG only
    assert got == want


LARGE_GAP_CASES = [(["-G", "1300"], 1250, 0, 3200), (["-G", "1300", "-GEC", "1", "-GOC", "2"], 1250, 0, 2000), (["-G", "450", "-MD", "200"], 420, 150, 1500), (["-G", "3000", "-MD", "400", "-BW", "8"], 2900, 300, 3500)]


@pytest.mark.skipif(not oracle.have_reference(), reason="oracle/_ref/yaha not built")
@pytest.mark.parametrize("extra,max_del,junk,flank", LARGE_GAP_CASES)
def test_large_gap_parameters_match_the_live_reference(work, index11, tmp_path, extra, max_del, junk, flank):
    # -G far beyond the goldens' 80: gap fills whose DP strip is hundreds of columns wide (the wave kernels' generic path on the device).  With the
    # default gap costs scoreClump splits such an alignment again (a 1 000-base deletion costs 2 005); with -GEC 0 it survives as one 1 000D op.
    import re
    from problems import write_long_indel_reads
    reads = str(tmp_path / "indel.fa")
    write_long_indel_reads(os.path.join(work, "genome_small.fa"), reads, 40, 11, max_del=max_del, junk=junk, flank=flank)
    ref_out = str(tmp_path / "ref.sam")
    oracle.run_reference(["-x", index11, "-q", reads, "-osh", ref_out] + extra)
    want = strip_pg(open(ref_out).read())
    with ya.Session(["-x", index11, "-q", reads] + extra) as s:
        b = s.next_batch(100)
        r, _own = oracle.run(s.index, s.params, b, threads=4)
        assert r.counters.dp_gap_cells > 40 * 4000                  # the wide gap fills really ran (strip width ~ the indel's length)
    assert run_oracle_pipeline(index11, reads, "-osh", extra) == want


@pytest.mark.parametrize("name", ["r1k", "rchim"])
def test_oracle_stages_match_the_instrumented_reference(work, index11, name):
    # fragment arrays after findFragmentsSort and every DP call of the REAL reference (an instrumented build, tests/golden/make_stage_golden.py)
    from problems import load_stage_golden, fasta_ids
    reads = os.path.join(work, name + ".fa")
    ids = fasta_ids(reads, 24)
    d, frags, probs, exp = load_stage_golden(name, ids)
    assert len(probs) > 400 and len(frags) >= 40
    with ya.Session(["-x", index11, "-q", reads]) as s:
        b = s.next_batch(24)
        got = {}
        for sro, sqo, eqo, rl, rs in oracle.seed_join(s.index, s.params, b):
            got.setdefault(rs, []).append((sro, sqo, eqo, rl))
        assert got == frags
        res = oracle.dp_batch(s.index, s.params, b, probs)
        bad = [k for k in range(len(probs)) if res[k] != exp[k]]
        assert not bad, "%d of %d DP calls differ from the reference, first: %r got %r exp %r" % (len(bad), len(probs), (probs[bad[0]].mode, probs[bad[0]].rOff, probs[bad[0]].qOff, probs[bad[0]].qLen), res[bad[0]], exp[bad[0]])


# ---- real human sequence (tests/golden/human, made by tests/golden/make_human_golden.py with the real reference binary at -L 15) -----------------------------
HUMAN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "human")


def unpack_human(d):
    """The pseudo-reference (1 000 hg18 reads of 10 kbp from the reference's testdata, concatenated), the two real read sets, and the simulated set (regenerated
    from its seed by this repo's simulator).  Returns the directory's file names."""
    import gzip, json, shutil, subprocess
    from conftest import ROOT
    for f in ("hs_pseudo.fa", "hs_real1k.fa", "hs_real200.fa"):
        with gzip.open(os.path.join(HUMAN, f + ".gz"), "rb") as g, open(os.path.join(d, f), "wb") as o:
            shutil.copyfileobj(g, o)
    meta = json.load(open(os.path.join(HUMAN, "human.json")))
    subprocess.check_call([os.path.join(ROOT, "tools", "yaha_sim"), "reads", "--genome", os.path.join(d, "hs_pseudo.fa"), "--out", os.path.join(d, "hs_sim1k.fa")] + meta["sim_args"])
    return meta


def human_golden(name):
    import gzip
    with gzip.open(os.path.join(HUMAN, name + ".out.gz"), "rb") as g:
        return g.read().decode().split("\n")


def test_real_human_sequence_at_the_default_seed_length(work, tmp_path):
    """Oracle + host stages == the real reference on human sequence: a 10 Mbp pseudo-reference of real hg18 reads (Alu / L1 / satellite content), real 1 kbp and
    200 bp hg18 reads that hit it through its repeat families, and reads sampled from it -- the reference's default -L 15 index (a 4.3 GB table), built by this
    repo's host builder."""
    d = str(tmp_path)
    meta = unpack_human(d)
    ya.build_index(["-g", os.path.join(d, "hs_pseudo.fa"), "-L", "15", "-cpuindex"])
    idx = os.path.join(d, "hs_pseudo.X15_01_65525S")
    for name, run in sorted(meta["runs"].items()):
        mine = run_oracle_pipeline(idx, os.path.join(d, run["reads"]), "-osh", [], batch=500, threads=8)
        assert mine == human_golden(name), "oracle + host stages differ from the reference on " + name
    os.remove(idx)
