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
