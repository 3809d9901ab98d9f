"""GPU tier, BASELINE-sized inputs: properties that do not need the oracle (it would take minutes at this size).

A 40 Mbp repeat-rich genome with the reference's defaults (-L 15) and 8 192 x 1 kbp reads (the bench workload's shape) go
through the whole device path; then, for every clump that comes back:
  * the edit list re-plays against the actual sequences: every base of an 'M' run matches, every base of an 'R' run differs,
    query span = M+R+I, reference span = M+R+D;
  * the clump's counters and its affine-gap score are those of its edit list (scoreClump, AlignHelpers.c:302-366);
  * reads are independent: the same reads in one batch, in two halves, and twice in a row give identical results;
  * a sub-sample small enough for the oracle is bit-exact against it.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
import yaha_amd as ya
from conftest import ROOT

pytestmark = pytest.mark.gpu

COMP = np.array([2, 3, 0, 1, 4, 12, 7, 6, 9, 8, 15, 11, 5, 13, 14, 10], dtype=np.uint8)     # fourBitCompCodes, Math.c:156


@pytest.fixture(scope="module")
def big(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("big"))
    sim = os.path.join(ROOT, "tools", "yaha_sim")
    g = os.path.join(d, "g.fa")
    subprocess.check_call([sim, "genome", "--seed", "7", "--out", g, "--seqs", "8", "--len", "5000000", "--repeat-frac", "0.45"])
    ya.build_index(["-g", g, "-L", "15"])
    reads = os.path.join(d, "r.fa")
    subprocess.check_call([sim, "reads", "--genome", g, "--out", reads, "--seed", "5", "--n", "8192", "--len", "1000", "--div", "0.017", "--chimeric", "0.05"])
    return os.path.join(d, "g.X15_01_65525S"), reads


def run_batches(index, reads, sizes):
    """Run the reads through the device in batches of the given sizes; returns per-read records + sequences."""
    recs, seqs = [], []
    with ya.Session(["-x", index, "-q", reads]) as s:
        P, ix = s.params, s.index
        nb = int(ix.n_base_bytes)
        packed = np.frombuffer((C.c_uint8 * nb).from_address(ix.bases), dtype=np.uint8)
        bases = np.empty(2 * nb, dtype=np.uint8); bases[0::2] = packed >> 4; bases[1::2] = packed & 15      # one code per reference offset
        with ya.Context(ix, P) as ctx:
            for n in sizes:
                b = s.next_batch(n)
                if b.n_reads == 0:
                    break
                offs = np.frombuffer((C.c_uint64 * (b.n_reads + 1)).from_address(b.offsets), dtype=np.uint64).astype(np.int64)
                codes = np.frombuffer((C.c_uint8 * int(offs[-1] - offs[0])).from_address(b.codes), dtype=np.uint8).copy()
                ctx.upload(b)
                ctx.run()
                r = ctx.collect()
                recs.extend(ya.result_records(r))
                for i in range(b.n_reads):
                    seqs.append(codes[offs[i] - offs[0]:offs[i + 1] - offs[0]])
        params = {k: getattr(P, k) for k in ("GOCost", "GECost", "RCost", "MScore", "minRawScore")}
    return recs, seqs, bases, params


def test_edit_lists_replay_and_reads_are_independent(big):
    index, reads = big
    recs, seqs, bases, P = run_batches(index, reads, [8192])
    assert len(recs) == 8192
    nclumps = nops = 0
    for rec, fwd in list(zip(recs, seqs))[:2000]:              # the replay is a Python loop: 2 000 reads, ~150 k clumps
        rev = COMP[fwd[::-1] & 15]
        for (sro, sqo, eqo, refLen, totScore, totLength, matched, mism, gap, status, ops) in rec:
            q = rev if (status & 1) else fwd
            qi, ri = sqo, sro
            m = r_ = ins = dele = 0
            ags = 0
            for ln, code in ops:
                if code in "MR":
                    qs = q[qi:qi + ln]
                    rs = bases[ri:ri + ln]
                    if code == "M":
                        assert (qs == rs).all(), "an M run holds a mismatch"
                        m += ln; ags += P["MScore"] * ln
                    else:
                        assert (qs != rs).all(), "an R run holds a match"
                        r_ += ln; ags -= P["RCost"] * ln
                    qi += ln; ri += ln
                elif code == "I":
                    ins += ln; qi += ln; ags -= P["GOCost"] + P["GECost"] * ln
                else:
                    assert code == "D"
                    dele += ln; ri += ln; ags -= P["GOCost"] + P["GECost"] * ln
            assert qi - sqo == eqo - sqo + 1 and ri - sro == refLen
            assert (matched, mism, gap, totLength) == (m, r_, ins + dele, m + r_ + ins + dele)
            assert totScore == ags & 0xFFFF and m >= P["minRawScore"] and (status & 0x08)
            nclumps += 1; nops += len(ops)
    assert nclumps > 2000 * 10 and nops > nclumps           # the genome is repeat-rich: many clumps per read
    # reads are independent units: any batching gives the same per-read results; a rerun is identical
    halves, _, _, _ = run_batches(index, reads, [4096, 4096])
    assert halves == recs
    ragged, _, _, _ = run_batches(index, reads, [1, 63, 1000, 3000, 4128])
    assert ragged == recs


def test_subsample_is_bit_exact_against_the_oracle(big):
    index, reads = big
    with ya.Session(["-x", index, "-q", reads]) as s:
        b = s.next_batch(300)
        with ya.Context(s.index, s.params) as ctx:
            ctx.upload(b)
            ctx.run()
            r = ctx.collect()
            ro, _own = oracle.run(s.index, s.params, b, threads=16)
            assert ya.result_records(r) == ya.result_records(ro)
            got, exp = r.counters.as_dict(), ro.counters.as_dict()
            for key in ("kmer_lookups", "hits", "fragments", "regions", "clumps_formed", "clumps_scored", "dp_ext_calls", "dp_gap_calls", "splits", "ops_out", "perfect_ext_bases"):
                assert got[key] == exp[key], (key, got[key], exp[key])


@pytest.mark.gpu
def test_state_words_are_checked_after_every_run(work, index11):
    """YGPU_CHECK_STATE=1 (on for the whole tier, tests/conftest.py): "the look-back and bucket words clean themselves up" is verified on the device at the end of
    every ygpu_run and ygpu_postfilter.  Here the check itself is checked, in a process of its own: a run is clean, a run that leaves one word dirty
    (YGPU_CHECK_STATE_INJECT) fails with YGPU_EINTERNAL and names the buffer -- and the run after it is clean again (the words are re-zeroed on failure)."""
    code = r"""
import os, sys
sys.path.insert(0, %r)
import yaha_amd as ya
with ya.Session(["-x", %r, "-q", %r]) as s:
    b = s.next_batch(48)
    with ya.Context(s.index, s.params) as c:
        c.upload(b); c.run(); first = ya.result_records(c.collect())
        c.set_postfilter(s); c.postfilter()
        os.environ["YGPU_CHECK_STATE_INJECT"] = "1"
        try:
            c.run(); print("NOT DETECTED")
        except Exception as e:
            print("DETECTED:", str(e)[:300])
        del os.environ["YGPU_CHECK_STATE_INJECT"]
        c.upload(b); c.run(); print("SAME" if ya.result_records(c.collect()) == first else "DIFFERENT")
""" % (ROOT, index11, os.path.join(work, "r1k.fa"))
    env = dict(os.environ, YGPU_CHECK_STATE="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600).stdout.decode()
    assert "DETECTED:" in out and "YGPU_CHECK_STATE" in out and "look-back" in out, out
    assert "NOT DETECTED" not in out and out.strip().endswith("SAME"), out
