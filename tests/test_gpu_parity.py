"""GPU tier: the HIP hot path (through the C-ABI, include/yaha_hip.h) against the oracle, stage by stage and
end to end, bit-exact; and the `yaha` CLI against SAM produced by the real reference (tests/golden)."""
import ctypes as C
import os
import subprocess

import pytest

import oracle
import yaha_amd as ya
from conftest import golden_lines, strip_pg, oflag_args, ROOT
from problems import dp_problems_from_chain

pytestmark = pytest.mark.gpu


def frag_tuples(f, n):
    return [(f[i].startRefOff, f[i].startQueryOff, f[i].endQueryOff, f[i].refLen, f[i].read_strand) for i in range(n)]


def _dp_check(ctx, probs, exp, kernels):
    res, ops, nops = ctx.dp_batch(probs, kernels)
    bad = 0
    for k, e in enumerate(exp):
        r = res[k]
        got = (r.score, r.addedQLen, r.addedRLen, tuple((ops[r.op_start + j] & 0xFFFF, chr((ops[r.op_start + j] >> 16) & 0xFF)) for j in range(r.n_ops)))
        if got != e:
            bad += 1
            if bad <= 3:
                p = probs[k]
                print("MISMATCH kernels", kernels, (p.read, p.strand, p.mode, p.qOff, p.qLen, p.rLen, p.rOff), "\n got", got, "\n exp", e)
    assert bad == 0, "%d of %d DP problems differ (kernel family %d)" % (bad, len(probs), kernels)


# Every findAffineGapScore call (SW.cpp:798-1208, through findAGSAlignment[Banded] / findAGS{Forward,Backward}Extension, SW.cpp:462-533), one by one,
# through EACH kernel family that can compute it: the wave-per-problem DP (dp_wave.h: every band width), and -- at the default band, where ygpu_run uses
# them for all DP cells -- the lane kernels: k_ext_rows + k_ext_trace, the careful-extension instantiation of k_ext_rows, the pure-diagonal shortcut and
# k_gap_lanes<16|32> / k_gap_wave.  -G 12 / -I 11 make the run caps bind inside the 21-column strip (k_ext_rows<CAPS>).
@pytest.mark.parametrize("reads,extra", [("r1k.fa", []), ("rchim.fa", ["-GOC", "0", "-GEC", "1"]), ("rq.fq", ["-BW", "7", "-X", "40"]), ("r10k.fa", ["-G", "30", "-BW", "12"]),
                                         ("r10k.fa", []), ("rchim.fa", ["-G", "12", "-I", "11"]), ("r1k.fa", ["-RC", "1", "-GOC", "9", "-GEC", "3", "-X", "9"])])
def test_dp_batch_bit_exact(work, index11, reads, extra):
    with ya.Session(["-x", index11, "-q", os.path.join(work, reads)] + extra) as s:
        b = s.next_batch(200)
        probs = dp_problems_from_chain(s, b, limit=4000, seed=5)
        exp = oracle.dp_batch(s.index, s.params, b, probs)
        lanes = s.params.bandWidth == 5 and s.params.maxGap >= 10
        with ya.Context(s.index, s.params) as ctx:
            ctx.upload(b)
            _dp_check(ctx, probs, exp, ya.DP_KERNELS_WAVE)
            _dp_check(ctx, probs, exp, ya.DP_KERNELS_AUTO)
            if lanes:
                _dp_check(ctx, probs, exp, ya.DP_KERNELS_LANES)
                _dp_check(ctx, probs, exp, ya.DP_KERNELS_LANES_CAREFUL)
            else:
                with pytest.raises(RuntimeError):
                    ctx.dp_batch(probs[:4], ya.DP_KERNELS_LANES)


@pytest.mark.parametrize("reads,extra", [("r1k.fa", []), ("rchim.fa", ["-H", "20"]), ("r100.fa", []), ("r10k.fa", [])])
def test_seed_join_and_chain_bit_exact(work, index11, reads, extra):
    with ya.Session(["-x", index11, "-q", os.path.join(work, reads)] + extra) as s:
        b = s.next_batch(500)
        with ya.Context(s.index, s.params) as ctx:
            ctx.upload(b)
            f, n = ctx.seed_join()
            assert frag_tuples(f, n) == oracle.seed_join(s.index, s.params, b)
            cf, cs, crs, nc = ctx.chain()
            got = [(crs[k], tuple((cf[i].startRefOff, cf[i].startQueryOff, cf[i].endQueryOff, cf[i].refLen) for i in range(cs[k], cs[k + 1]))) for k in range(nc)]
            assert got == oracle.chain(s.index, s.params, b)


def test_hits_either_side_of_long_runs_of_k_mers_without_hits(work, index11, tmp_path):
    """k_expand_hits stages the k-mers of a block of 1 024 hits in a window of 1 026 offsets; a run of k-mers without hits longer than that (reads of N,
    reads that match nowhere) between two k-mers with hits sends the hits behind it through the bisection path.  Fragments and chains against the oracle."""
    import random
    rng = random.Random(77)
    src = [l.rstrip("\n") for l in open(os.path.join(work, "r1k.fa"))]
    recs, cur = [], None
    for l in src:
        if l.startswith(">"):
            cur = [l, ""]; recs.append(cur)
        else:
            cur[1] += l
    out = []
    for k, (h, seq) in enumerate(recs[:120]):
        out.append((h, seq))
        if k % 3 == 0:
            out.append((">allN_%d" % k, "N" * rng.choice((1100, 2500, 4000))))
        if k % 7 == 0:
            out.append((">poly_%d" % k, "".join(rng.choice("ACGT") for _ in range(40)) + "N" * 1500 + seq[:300]))
    path = os.path.join(str(tmp_path), "mixed.fa")
    with open(path, "w") as f:
        for h, seq in out:
            f.write(h + "\n" + seq + "\n")
    with ya.Session(["-x", index11, "-q", path]) as s:
        b = s.next_batch(1000)
        assert b.n_reads == len(out)
        with ya.Context(s.index, s.params) as ctx:
            ctx.upload(b)
            f, n = ctx.seed_join()
            assert frag_tuples(f, n) == oracle.seed_join(s.index, s.params, b)
            cf, cs, crs, nc = ctx.chain()
            got = [(crs[k], tuple((cf[i].startRefOff, cf[i].startQueryOff, cf[i].endQueryOff, cf[i].refLen) for i in range(cs[k], cs[k + 1]))) for k in range(nc)]
            assert got == oracle.chain(s.index, s.params, b)
            ctx.run()
            r = ctx.collect()
            ro, _own = oracle.run(s.index, s.params, b, threads=8)
            assert ya.result_records(r) == ya.result_records(ro)


def device_pipeline(index, reads, oflag, extra, batch=4096):
    out = []
    with ya.Session(["-x", index, "-q", reads] + oflag_args(oflag) + list(extra)) as s:
        out.append(s.header())
        with ya.Context(s.index, s.params) as ctx:
            while True:
                b = s.next_batch(batch)
                if b.n_reads == 0:
                    break
                ctx.upload(b)
                ctx.run()
                r = ctx.collect()
                ro, _own = oracle.run(s.index, s.params, b, threads=8)
                assert ya.result_records(r) == ya.result_records(ro), "device clump records differ from the oracle"
                got, exp = r.counters.as_dict(), ro.counters.as_dict()
                for key in ("kmer_lookups", "hits", "fragments", "regions", "clumps_formed", "clumps_scored", "dp_ext_calls", "dp_gap_calls", "splits", "ops_out", "perfect_ext_bases"):
                    assert got[key] == exp[key], (key, got[key], exp[key])
                out.append(s.emit(r))
    return strip_pg("".join(out))


def test_every_golden_run_matches_on_the_device(work, index11, meta):
    for name, run in sorted(meta["runs"].items()):
        mine = device_pipeline(index11, os.path.join(work, run["reads"]), run["oflag"], run["extra"], batch=130)
        assert mine == golden_lines(name), "HIP path differs from the reference on " + name


# Parameter corners of the lane pipeline (-BW 5): run caps that bind inside the 21-column strip (k_ext_rows<true>), maxGap just
# at the lane kernels' limit, scoring where the pure-diagonal shortcut of phase 1 is wide (cheap mismatches) or never applies
# (expensive mismatches), a tight X-drop.  device_pipeline() asserts device == oracle per batch (records and work counters);
# when the reference binary travelled with the snapshot its SAM is the referee as well.
@pytest.mark.parametrize("reads,extra", [
    ("r1k.fa", ["-G", "12"]), ("rchim.fa", ["-G", "20", "-I", "11"]), ("r10k.fa", ["-G", "10"]), ("r1k.fa", ["-G", "9"]),
    ("rchim.fa", ["-RC", "1", "-GOC", "9", "-GEC", "3"]), ("r1k.fa", ["-RC", "9", "-GOC", "1", "-GEC", "1"]), ("r10k.fa", ["-X", "6", "-MS", "2"]),
    ("rq.fq", ["-G", "21", "-I", "20"])])
def test_lane_pipeline_parameter_corners(work, index11, reads, extra, tmp_path):
    mine = device_pipeline(index11, os.path.join(work, reads), "-osh", extra, batch=700)
    if oracle.have_reference() and "-I" not in extra:          # -I exists in experimental builds of the reference only
        ref_out = str(tmp_path / "ref.sam")
        oracle.run_reference(["-x", index11, "-q", os.path.join(work, reads), "-osh", ref_out] + list(extra))
        assert mine == strip_pg(open(ref_out).read())


# The post-filter on the device (ygpu_postfilter: OQC, filter by similarity, mapping quality -- oqc_core.h, one read per lane) against the host's filter over the
# same hot-path results: the SAM text of what the device returns (only the clumps that are printed) must equal the text the host makes from everything.
# The parameter sets move every knob of the filter: -FBS with loose and tight similarity, break point cost and its log cap (the step-function table), the
# minimum non-overlap, soft clipping, scoring that changes the overlap scores, and chimeric / long / FASTQ reads for paths of several primaries.
@pytest.mark.parametrize("reads,extra", [
    ("r1k.fa", []), ("rchim.fa", []), ("rchim.fa", ["-FBS", "Y"]), ("rchim.fa", ["-FBS", "Y", "-PRL", "0.5", "-PSS", "0.5"]), ("rchim.fa", ["-BP", "0"]), ("rchim.fa", ["-BP", "17", "-MGDP", "9"]),
    ("rchim.fa", ["-MGDP", "1", "-MNO", "3"]), ("rchim.fa", ["-MNO", "200", "-FBS", "Y", "-PSS", "0.2", "-PRL", "0.1"]), ("r10k.fa", ["-FBS", "Y"]), ("rq.fq", ["-M", "15", "-P", "0.8", "-FBS", "Y", "-PSS", "0.3"]),
    ("rchim.fa", ["-GOC", "9", "-GEC", "3", "-RC", "1", "-FBS", "Y"]), ("r100.fa", ["-M", "15", "-FBS", "Y", "-PRL", "0.3", "-PSS", "0.3"])])
def test_postfilter_on_the_device_equals_the_host_filter(work, index11, reads, extra, monkeypatch):
    """Two compilations of ONE routine (csrc/oqc_core.h) against each other: the device stage vs the host's copy, over the same hot-path results.  The pin to the
    REFERENCE is indirect and lives elsewhere: the host's copy == the reference's SAM in every golden of the CPU tier (test_oracle_golden.py, Session.emit) and in the
    live-reference comparisons; the device stage == the reference's SAM in the command-line tests of this tier (the device filter is the CLI's default:
    test_cli_drop_in*, test_real_human_sequence*, test_gpu_long_reads.py, test_gpu_at_scale.py)."""
    if reads == "rchim.fa" and not extra:
        monkeypatch.setenv("YGPU_OQC_MAX", "4")             # reads of more than four clumps take the hand-over path (unfiltered, marked, filtered by the host)
    with ya.Session(["-x", index11, "-q", os.path.join(work, reads), "-osh", "stdout"] + list(extra)) as s:
        with ya.Context(s.index, s.params) as ctx:
            ctx.set_postfilter(s)
            total = 0
            while True:
                b = s.next_batch(173)
                if b.n_reads == 0:
                    break
                ctx.upload(b); ctx.run()
                host = s.emit(ctx.collect())
                f = ctx.postfilter()
                assert f.n_reads == b.n_reads
                dev = s.emit_filtered(f)
                assert dev == host
                total += len(host)
            assert total > 0


# Bands wider than a wave has lanes for (4 * BW + 1 > 64 columns in the X-drop extension: -BW 16 and up) run through the sequential recurrence of dp_wave.h;
# the reference accepts any band (Main.c:324-327).  device == oracle per batch, and the reference binary's SAM when it is there.
@pytest.mark.parametrize("reads,extra", [("r1k.fa", ["-BW", "16"]), ("rchim.fa", ["-BW", "20", "-G", "80"]), ("r10k.fa", ["-BW", "40"]), ("rq.fq", ["-BW", "33", "-X", "40"])])
def test_bands_wider_than_a_wave(work, index11, reads, extra, tmp_path):
    mine = device_pipeline(index11, os.path.join(work, reads), "-osh", extra, batch=300)
    if oracle.have_reference():
        ref_out = str(tmp_path / "ref.sam")
        oracle.run_reference(["-x", index11, "-q", os.path.join(work, reads), "-osh", ref_out] + list(extra))
        assert mine == strip_pg(open(ref_out).read())


@pytest.mark.parametrize("budget", ["3000", "40000"])
def test_trace_memory_chunks(work, index11, meta, budget, monkeypatch):
    # When the extension trace strips of a batch do not fit in device memory the roots are processed in chunks that reuse one
    # buffer (long reads, big batches).  YGPU_TRACE_BUDGET_BLOCKS (read at ygpu_init) forces that path on small inputs.
    monkeypatch.setenv("YGPU_TRACE_BUDGET_BLOCKS", budget)
    for name in ("r10k_default", "rchim_default", "r1k_default"):
        run = meta["runs"][name]
        mine = device_pipeline(index11, os.path.join(work, run["reads"]), run["oflag"], run["extra"], batch=400)
        assert mine == golden_lines(name), "chunked trace path differs from the reference on " + name


@pytest.mark.parametrize("mode", [("2", "16384"), ("2", "700"), ("2", "3000"), ("2", "64"), ("2", "16384", "0"), ("2", "16384", "1", "ballots"), ("2", "700", "0", "ballots"),
                                  ("2", "3000", "1", "atomic")])
def test_hit_sort_paths(work, index11, meta, mode, monkeypatch):
    # A2's sort: one workgroup per (read, strand) in size classes; segments above YGPU_SEGSORT_MAX hits are cut by diagonal, and a piece that still does not fit is cut
    # again over its own range of diagonals until it fits or holds one diagonal -- the limit is lowered here so that the 10 kbp reads' segments take every class, the
    # long-segment path and (64: a true alignment's thousands of hits on one diagonal) several levels of cuts down to single-diagonal pieces.
    monkeypatch.setenv("YGPU_SEG_SORT", mode[0])
    if mode[1]:
        monkeypatch.setenv("YGPU_SEGSORT_MAX", mode[1])
    if len(mode) > 2:
        monkeypatch.setenv("YGPU_SORT_WIDE", mode[2])         # 0: the four largest classes in their 1 024-thread shapes (the default since round 5: 512 threads, twice the hits a thread)
    if len(mode) > 3:
        monkeypatch.setenv("YGPU_SORT_RANK", mode[3])         # the workgroup sort's ranking (wgsort.h): LDS atomics (the default; "atomic": no fallback) or ballots
    for name in ("r10k_default", "r1k_default"):
        run = meta["runs"][name]
        mine = device_pipeline(index11, os.path.join(work, run["reads"]), run["oflag"], run["extra"], batch=400)
        assert mine == golden_lines(name), "hit sort path %r differs from the reference on %s" % (mode, name)


def test_two_contexts_in_flight_match_one(work, index11):
    # ygpu_clone: a second context on the same device sharing the index image; two host threads step them concurrently
    import threading
    with ya.Session(["-x", index11, "-q", os.path.join(work, "r10k.fa")]) as s1, ya.Session(["-x", index11, "-q", os.path.join(work, "rchim.fa")]) as s2:
        b1, b2 = s1.next_batch(400), s2.next_batch(400)
        with ya.Context(s1.index, s1.params) as solo:
            solo.upload(b1); solo.run(); exp1 = ya.result_records(solo.collect())
            solo.upload(b2); solo.run(); exp2 = ya.result_records(solo.collect())
        with ya.Context(s1.index, s1.params) as a:
            b = ya.Context(s1.index, s1.params, parent=a)
            got = {}

            def work_on(ctx, batch, key):
                for _ in range(3):
                    ctx.upload(batch); ctx.run(); got[key] = ya.result_records(ctx.collect())
            th = [threading.Thread(target=work_on, args=(a, b1, 1)), threading.Thread(target=work_on, args=(b, b2, 2))]
            for x in th:
                x.start()
            for x in th:
                x.join()
            b.close()
        assert got[1] == exp1 and got[2] == exp2


def test_small_batch_then_large_batch_on_one_context(work, index11, monkeypatch):
    # after its first batch a context bounds the clump slots by what the last batch used (the slots are pre-set, a store each); a batch that overflows the
    # bound is redone at the full one and must give the records a fresh context gives.  YGPU_CLUMP_BOUND sets a first bound that the batch overflows.
    with ya.Session(["-x", index11, "-q", os.path.join(work, "r1k.fa")]) as s, ya.Session(["-x", index11, "-q", os.path.join(work, "r1k.fa")]) as s0:
        small = s0.next_batch(3)                                         # (a session's batches share its buffers: one session per batch kept)
        large = s.next_batch(600)
        with ya.Context(s.index, s.params) as fresh:
            fresh.upload(large); fresh.run(); exp = ya.result_records(fresh.collect())
        with ya.Context(s.index, s.params) as ctx:
            ctx.upload(small); ctx.run(); ya.result_records(ctx.collect())
            ctx.upload(large); ctx.run(); got = ya.result_records(ctx.collect())
            monkeypatch.setenv("YGPU_CLUMP_BOUND", "64")
            ctx.upload(large); ctx.run(); redone = ya.result_records(ctx.collect())
            monkeypatch.delenv("YGPU_CLUMP_BOUND")
            ctx.upload(small); ctx.run(); ctx.upload(large); ctx.run(); again = ya.result_records(ctx.collect())
        assert len(exp) > 100 and got == exp and redone == exp and again == exp


def test_cli_drop_in(work, index11, tmp_path):
    out = str(tmp_path / "o.sam")
    subprocess.check_call([ya.CLI_PATH, "-x", index11, "-q", os.path.join(work, "rchim.fa"), "-osh", out, "-FBS", "Y", "-t", "4", "-batch", "64"], stderr=subprocess.DEVNULL)
    mine = strip_pg(open(out).read())
    ref = golden_lines("rchim_FBS")
    # the golden was written with -oss; redo with the matching flag
    subprocess.check_call([ya.CLI_PATH, "-x", index11, "-q", os.path.join(work, "rchim.fa"), "-oss", out, "-FBS", "Y"], stderr=subprocess.DEVNULL)
    assert strip_pg(open(out).read()) == ref
    # one context per GPU and three give the same output as the default two (batch tickets order the output)
    for nctx in ("1", "3"):
        subprocess.check_call([ya.CLI_PATH, "-x", index11, "-q", os.path.join(work, "rchim.fa"), "-oss", out, "-FBS", "Y", "-ctx", nctx, "-batch", "50"], stderr=subprocess.DEVNULL)
        assert strip_pg(open(out).read()) == ref
    assert len(mine) == len(ref)



@pytest.mark.gpu
def test_cli_on_a_version_1_nib2(work, tmp_path):
    """The command line on a `.nib2` of VERSION 1 (12-byte sequence entries, 16-bit name fields: Compress.c:89-128, host/formats.cpp parseNib2) gives the SAM of the
    version-2 file -- the reference's golden."""
    from conftest import nib2_v1_copy
    idx1 = nib2_v1_copy(work, str(tmp_path / "v1")); out = str(tmp_path / "o.sam")
    subprocess.check_call([ya.CLI_PATH, "-x", idx1, "-q", os.path.join(work, "rchim.fa"), "-osh", out], stderr=subprocess.DEVNULL)
    assert strip_pg(open(out).read()) == golden_lines("rchim_default")


def test_a_failed_order_check_behind_the_sort_falls_back_to_the_ballots(work, index11, tmp_path):
    """The workgroup sort ranks with LDS atomics, whose lane order no manual promises; k_frag_scan_build checks every batch's keys against their predecessors, and a
    batch that fails is sorted again with the ranking by ballots, which the process then keeps (stage_seed.hip).  YGPU_SORT_CHECK_INJECT=1 fails the first check: the
    run says so once, and its SAM is the reference's; with YGPU_SORT_RANK=atomic there is no fallback, and nothing is injected."""
    out = str(tmp_path / "o.sam"); ref = golden_lines("rchim_FBS")
    cmd = [ya.CLI_PATH, "-x", index11, "-q", os.path.join(work, "rchim.fa"), "-oss", out, "-FBS", "Y", "-ctx", "2", "-batch", "50"]
    p = subprocess.run(cmd, stderr=subprocess.PIPE, env=dict(os.environ, YGPU_SORT_CHECK_INJECT="1"))
    assert p.returncode == 0, p.stderr.decode()[-1500:]
    assert p.stderr.decode().count("stays with the ranking by ballots") == 1
    assert strip_pg(open(out).read()) == ref
    p = subprocess.run(cmd, stderr=subprocess.PIPE, env=dict(os.environ, YGPU_SORT_CHECK_INJECT="1", YGPU_SORT_RANK="atomic"))
    assert p.returncode == 0 and b"ballots" not in p.stderr and strip_pg(open(out).read()) == ref
    p = subprocess.run(cmd, stderr=subprocess.PIPE)
    assert p.returncode == 0 and b"ballots" not in p.stderr and strip_pg(open(out).read()) == ref


def test_cli_small_batches_many_times(work, index11, tmp_path):
    """Thirty runs of the command line over batches of 50 reads on three contexts -- every context presized from the first, dozens of hand-overs between context and
    filter threads a run: the same SAM every time.  (Round 5: a presized context's post-filter started on unzeroed look-back words -- one run in fifty ended in a
    device fault, another few in a different SAM with exit code 0; tools/cli_stress.sh is the longer form of this test.)"""
    out = str(tmp_path / "o.sam"); ref = golden_lines("rchim_FBS")
    for i in range(30):
        p = subprocess.run([ya.CLI_PATH, "-x", index11, "-q", os.path.join(work, "rchim.fa"), "-oss", out, "-FBS", "Y", "-ctx", "3", "-batch", "50"], stderr=subprocess.PIPE)
        assert p.returncode == 0, (i, p.stderr.decode()[-1500:])
        assert strip_pg(open(out).read()) == ref, "run %d differs" % i


@pytest.mark.skipif(not oracle.have_reference(), reason="oracle/_ref/yaha not present")
def test_live_reference_binary_on_fresh_human_like_reads(work, tmp_path):
    # bigger, repeat-richer genome built on the box; the reference binary itself is the referee
    sim = os.path.join(ROOT, "tools", "yaha_sim")
    g = str(tmp_path / "g.fa")
    subprocess.check_call([sim, "genome", "--seed", "99", "--out", g, "--seqs", "4", "--len", "1500000", "--repeat-frac", "0.5"])
    ya.build_index(["-g", g, "-L", "12"])
    idx = str(tmp_path / "g.X12_01_65525S")
    reads = str(tmp_path / "r.fa")
    subprocess.check_call([sim, "reads", "--genome", g, "--out", reads, "--seed", "3", "--n", "1500", "--len", "1000", "--div", "0.017", "--chimeric", "0.1"])
    ref_out = str(tmp_path / "ref.sam")
    oracle.run_reference(["-x", idx, "-q", reads, "-osh", ref_out])
    out = str(tmp_path / "mine.sam")
    subprocess.check_call([ya.CLI_PATH, "-x", idx, "-q", reads, "-osh", out, "-t", "4"], stderr=subprocess.DEVNULL)
    assert strip_pg(open(out).read()) == strip_pg(open(ref_out).read())


# -G / -MD far beyond the goldens: gap fills whose strip is hundreds of columns wide and rows x columns exceeds the extension strip's footprint -- the wave
# kernels' generic path with its scratch sized from the parameters (stage_align.hip alignDims).  The reference itself behaves oddly out there (16-bit fields
# overflow); the oracle reproduces it (CPU tier: test_large_gap_parameters_match_the_live_reference) and the device has to as well.
@pytest.mark.parametrize("extra,max_del,junk,flank", [(["-G", "1300"], 1250, 0, 3200), (["-G", "1300", "-GEC", "1", "-GOC", "2"], 1250, 0, 2000), (["-G", "450", "-MD", "200"], 420, 150, 1500),
                                                      (["-G", "3000", "-MD", "400", "-BW", "8"], 2900, 300, 3500)])
def test_large_gap_parameters(work, index11, tmp_path, extra, max_del, junk, flank):
    from problems import write_long_indel_reads
    reads = str(tmp_path / "indel.fa")
    write_long_indel_reads(os.path.join(work, "genome_small.fa"), reads, 40, 11, max_del=max_del, junk=junk, flank=flank)
    mine = device_pipeline(index11, reads, "-osh", extra, batch=64)
    if oracle.have_reference():
        ref_out = str(tmp_path / "ref.sam")
        oracle.run_reference(["-x", index11, "-q", reads, "-osh", ref_out] + list(extra))
        assert mine == strip_pg(open(ref_out).read())


def test_async_tickets_one_host_thread_two_contexts(work, index11):
    # ygpu_submit / ygpu_poll / ygpu_wait: this thread keeps two contexts busy; results equal the synchronous calls
    import time
    with ya.Session(["-x", index11, "-q", os.path.join(work, "r10k.fa")]) as s1, ya.Session(["-x", index11, "-q", os.path.join(work, "rchim.fa")]) as s2:
        b1, b2 = s1.next_batch(300), s2.next_batch(300)
        with ya.Context(s1.index, s1.params) as a:
            a.upload(b1); a.run(); exp1 = ya.result_records(a.collect())
            a.upload(b2); a.run(); exp2 = ya.result_records(a.collect())
            b = ya.Context(s1.index, s1.params, parent=a)
            for _ in range(2):
                t1, t2 = a.submit(b1), b.submit(b2)
                with pytest.raises(RuntimeError):
                    a.submit(b2)                                    # one open ticket per context
                while not (a.poll(t1) == 1 and b.poll(t2) == 1):
                    time.sleep(0.001)
                assert ya.result_records(a.wait(t1)) == exp1 and ya.result_records(b.wait(t2)) == exp2
            with pytest.raises(RuntimeError):
                a.wait(t1)                                          # the ticket is closed
            t = a.submit(b2)
            assert ya.result_records(a.wait(t)) == exp2             # wait without polling
            b.close()


def _sha(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def test_index_built_on_the_device_is_byte_identical(work, meta, tmp_path):
    # count -> scan -> fill -> order -> sample on the GPU (device/index_build.hip) against the reference's own files (SHA-256 goldens)
    # and against this repo's host builder (the reference's three passes restated), incl. the Floyd sampling of over-represented k-mers.
    import shutil
    d = str(tmp_path)
    shutil.copy(os.path.join(work, "genome_small.nib2"), os.path.join(d, "genome_small.nib2"))
    g = os.path.join(d, "genome_small.nib2")
    for args, name in ((["-L", "11"], "genome_small.X11_01_65525S"), (["-L", "8", "-H", "20"], "genome_small.X08_01_00020S")):
        r = subprocess.run([ya.CLI_PATH, "-g", g] + args, stderr=subprocess.PIPE, check=True)
        assert b"Building the index on GPU" in r.stderr
        assert os.path.getsize(os.path.join(d, name)) == meta["index"][name]["size"] if "size" in meta["index"][name] else True
        assert _sha(os.path.join(d, name)) == meta["index"][name]["sha256"], name
    # a repeat-rich 6 Mbp genome: -L 15 (the default; 4.3 GB table), and -L 10 -H 40 where thousands of k-mers are sampled and the long lists take
    # the workgroup and device sorts
    big = os.path.join(d, "big.fa")
    subprocess.check_call([os.path.join(ROOT, "tools", "yaha_sim"), "genome", "--seed", "5", "--out", big, "--seqs", "5", "--len", "6000000", "--repeat-frac", "0.5", "--nrun", "3", "--lowcomplex", "6"])
    for args, name in ((["-L", "15"], "big.X15_01_65525S"), (["-L", "10", "-H", "40"], "big.X10_01_00040S"), (["-L", "12", "-H", "3000"], "big.X12_01_03000S"),
                       (["-L", "13", "-S", "2"], "big.X13_02_65525S"), (["-L", "11", "-S", "7", "-H", "100"], "big.X11_07_00100S"), (["-L", "12", "-S", "12"], "big.X12_12_65525S")):      # -S > 1: starts follow the reference's scan (N runs re-phase it)
        subprocess.run([ya.CLI_PATH, "-g", big] + args, stderr=subprocess.DEVNULL, check=True)
        dev = os.path.join(d, name + ".device")
        os.rename(os.path.join(d, name), dev)
        r = subprocess.run([ya.CLI_PATH, "-g", big, "-cpuindex"] + args, stderr=subprocess.PIPE, check=True)
        assert b"Building the index on GPU" not in r.stderr
        assert subprocess.run(["cmp", "-s", dev, os.path.join(d, name)]).returncode == 0, "device-built index differs from the host-built one: " + name
        os.remove(dev)
        if "-H" in args:
            # once more with the lists of more than 64 entries handed to the multi-workgroup sort (a bitonic network over padded copies, index_build.hip): what only the few
            # satellite k-mers of a 3 Gbp genome take otherwise (a list beyond 8 192 entries)
            os.rename(os.path.join(d, name), os.path.join(d, name + ".host"))
            subprocess.run([ya.CLI_PATH, "-g", big] + args, stderr=subprocess.DEVNULL, check=True, env=dict(os.environ, YAHA_IX_HUGE_MIN="64"))
            assert subprocess.run(["cmp", "-s", os.path.join(d, name + ".host"), os.path.join(d, name)]).returncode == 0, "device-built index (long lists through the multi-workgroup sort) differs from the host-built one: " + name
            os.remove(os.path.join(d, name + ".host"))
        os.remove(os.path.join(d, name))


@pytest.mark.parametrize("reads,extra", [("r1k.fa", []), ("rchim.fa", []), ("r10k.fa", [])])
def test_both_extension_kernel_families(work, index11, reads, extra, monkeypatch):
    # The X-drop extensions run in packed 16-bit arithmetic (k_ext_rows_pk / k_ext_trace_pk, ext_lanes_pk.h) whenever the scores fit, which is every default
    # run up to 15 kbp reads; YGPU_EXT32 forces the 32-bit kernels (k_ext_rows / k_ext_trace) on the same input.  Whole pipeline (records and work counters
    # against the oracle in device_pipeline) and the stage entry for the lane kernels, both ways; the two SAM texts must be the same text.
    path = os.path.join(work, reads)
    packed = device_pipeline(index11, path, "-osh", extra, batch=150)
    with ya.Session(["-x", index11, "-q", path] + extra) as s:
        b = s.next_batch(200)
        probs = dp_problems_from_chain(s, b, limit=4000, seed=11)
        exp = oracle.dp_batch(s.index, s.params, b, probs)
        with ya.Context(s.index, s.params) as ctx:
            ctx.upload(b)
            _dp_check(ctx, probs, exp, ya.DP_KERNELS_LANES)
            monkeypatch.setenv("YGPU_EXT32", "1")
            _dp_check(ctx, probs, exp, ya.DP_KERNELS_LANES)
            _dp_check(ctx, probs, exp, ya.DP_KERNELS_LANES_CAREFUL)
    assert device_pipeline(index11, path, "-osh", extra, batch=150) == packed


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_packed_rows_random_scoring(work, index11, seed):
    # scoring parameters drawn at random inside the packed 16-bit kernel's limits (and the reference's own: GEC >= 1): the whole pipeline, records and work
    # counters, against the oracle; chimeric and plain reads, both extension directions, splits
    import random
    rnd = random.Random(1000 + seed)
    extra = ["-MS", str(rnd.randint(1, 4)), "-RC", str(rnd.randint(1, 9)), "-GOC", str(rnd.randint(1, 14)), "-GEC", str(rnd.randint(1, 6)), "-X", str(rnd.choice([8, 25, 60, 150, 400]))]
    for reads in ("r1k.fa", "rchim.fa"):
        assert device_pipeline(index11, os.path.join(work, reads), "-osh", extra, batch=200)


@pytest.mark.parametrize("xdrop", ["25", "3900", "4100"])
def test_packed_rows_at_the_ends_of_their_score_range(work, index11, tmp_path, xdrop):
    # k_ext_rows_pk computes in saturating 16-bit arithmetic with a sentinel of -16000 and is used while MS * (longest read) <= 15000 and
    # RC + X + GO + 21 GE <= 4000 (ext_lanes_pk.h).  Reads of 15000 bases (the last length it takes), 15001 (the first the 32-bit kernel gets) and 9000, each an
    # exact copy of the genome behind 1200 random bases: forward extensions that climb to a score of ~13800, backward extensions that stay below
    # zero for as long as the X-drop allows (-X 3900: the largest the packed kernel takes; 4100: the 32-bit kernel), every result against the oracle.
    import random
    from problems import read_fasta
    _, seqs = read_fasta(os.path.join(work, "genome_small.fa"))
    rnd = random.Random(7)
    for k, (total, start) in enumerate([(15000, 2000), (15001, 40000), (9000, 70000)]):
        junk = "".join(rnd.choice("ACGT") for _ in range(1200))
        reads = str(tmp_path / ("long%d.fa" % k))
        with open(reads, "w") as f:
            body = list(seqs[0][start:start + total - 1200])
            for pos in range(900, len(body), 2500):                              # a substitution every 2 500 bases: the seed fragments end there, the extensions run on
                body[pos] = "ACGT"[("ACGT".index(body[pos]) + 1) % 4]
            f.write(">long%d\n%s%s\n" % (k, junk, "".join(body)))
        with ya.Session(["-x", index11, "-q", reads, "-X", xdrop]) as s:
            b = s.next_batch(4)
            assert b.n_reads == 1
            probs = []
            for rs, frags in oracle.chain(s.index, s.params, b):
                f0 = frags[0]
                probs.append(ya.DPProblem(0, rs & 1, ya.DP_EXT_REV, f0[1] - 1, min(f0[1], f0[0]), 0, f0[0] - 1))
                probs.append(ya.DPProblem(0, rs & 1, ya.DP_EXT_FWD, f0[2] + 1, total - 1 - f0[2], 0, f0[0] + f0[3]))
            assert probs
            exp = oracle.dp_batch(s.index, s.params, b, probs)
            assert max(e[0] for e in exp) > total - 1200 - 1000 - 40                # the long climb is among them (5 substitutions, first fragment of 900 bases)
            with ya.Context(s.index, s.params) as ctx:
                ctx.upload(b)
                _dp_check(ctx, probs, exp, ya.DP_KERNELS_LANES)
                _dp_check(ctx, probs, exp, ya.DP_KERNELS_LANES_CAREFUL)
        assert device_pipeline(index11, reads, "-osh", ["-X", xdrop], batch=4)


@pytest.mark.parametrize("name", ["r1k", "rchim"])
def test_device_stages_match_the_instrumented_reference(work, index11, name):
    # ygpu_seed_join and ygpu_dp_batch (every kernel family) replayed against dumps of the REAL reference -- the fragment arrays after
    # findFragmentsSort and each DP call's arguments and results, printed by an instrumented build (tests/golden/make_stage_golden.py, fixtures
    # tests/golden/stage_*.json.gz) -- not against this repo's own restatement.
    from problems import load_stage_golden, fasta_ids
    reads = os.path.join(work, name + ".fa")
    ids = fasta_ids(reads, 24)
    d, frags, probs, exp = load_stage_golden(name, ids)
    with ya.Session(["-x", index11, "-q", reads]) as s:
        b = s.next_batch(24)
        with ya.Context(s.index, s.params) as ctx:
            ctx.upload(b)
            f, n = ctx.seed_join()
            got = {}
            for sro, sqo, eqo, rl, rs in frag_tuples(f, n):
                got.setdefault(rs, []).append((sro, sqo, eqo, rl))
            assert got == frags
            for kernels in (ya.DP_KERNELS_AUTO, ya.DP_KERNELS_WAVE, ya.DP_KERNELS_LANES, ya.DP_KERNELS_LANES_CAREFUL):
                _dp_check(ctx, probs, exp, kernels)


def _synthetic_results(s, b, rng, max_clumps):
    """Clump lists no real read produces, for the reads of batch b: hundreds of clumps a read, scores from a handful of values, exact duplicates (same span, same place, same
    strand), the same span at other places (copies of a repeat), nested and overlapping spans with edit lists of all four ops -- consistent records (query bases of the ops =
    the span, reference bases = refLen, everything inside the genome)."""
    import numpy as np
    offs = C.cast(b.offsets, C.POINTER(C.c_uint64))
    gmax = int(s.index.maxROff)
    starts, recs, ops = [0], [], []

    def make_ops(qspan):
        out, q, r = [], 0, 0
        while q < qspan:
            kind = "MMMMMRID"[int(rng.integers(0, 8))]
            ln = int(min(qspan - q, rng.integers(1, 40))) if kind != "D" else int(rng.integers(1, 6))
            if kind == "D" and (not out or q == 0):
                kind = "M"; ln = int(min(qspan - q, rng.integers(1, 40)))
            out.append((ln, kind)); q += ln if kind != "D" else 0; r += ln if kind in "MRD" else 0
        merged = []
        for ln, k in out:                               # adjacent ops of one kind are one op
            if merged and merged[-1][1] == k:
                merged[-1] = (merged[-1][0] + ln, k)
            else:
                merged.append((ln, k))
        r = sum(l for l, k in merged if k in "MRD")
        return merged, r

    for i in range(b.n_reads):
        qlen = int(offs[i + 1] - offs[i])
        n = int(rng.integers(0, max_clumps + 1)) if rng.random() < 0.9 else int(rng.integers(0, 4))
        mine = []
        while len(mine) < n:
            sqo = int(rng.integers(0, max(1, qlen - 30))); eqo = int(min(qlen - 1, sqo + rng.integers(24, max(25, qlen // int(rng.integers(1, 6))))))
            o, rlen = make_ops(eqo - sqo + 1)
            sro = int(rng.integers(1000, gmax - 40000)); score = int(rng.choice([25, 30, 30, 40, 60, 60, 100, 200, 400, 800])); rev = int(rng.random() < 0.5)
            rec = (sro, sqo, eqo, rlen, score, o, rev)
            mine.append(rec)
            for _ in range(int(rng.integers(0, 4))):      # copies: exact duplicates, or the same span elsewhere / on the other strand
                if len(mine) >= n:
                    break
                mine.append(rec if rng.random() < 0.4 else (int(rng.integers(1000, gmax - 40000)), sqo, eqo, rlen, score, o, int(rng.random() < 0.5)))
        order = rng.permutation(len(mine))
        for k in order:
            sro, sqo, eqo, rlen, score, o, rev = mine[k]
            recs.append((sro, sqo, eqo, rlen, score, eqo - sqo + 1, sum(l for l, c in o if c == "M"), sum(l for l, c in o if c == "R"), sum(l for l, c in o if c in "ID"), rev, 0, len(ops), len(o)))
            ops.extend(l | (ord(c) << 16) for l, c in o)
        starts.append(len(recs))
    cs = (C.c_uint32 * len(starts))(*starts); cl = (ya.Clump * max(1, len(recs)))(*[ya.Clump(*r) for r in recs]); op = (C.c_uint32 * max(1, len(ops)))(*ops)
    r = ya.ResultBatch(); r.n_reads = b.n_reads; r.clump_start = cs; r.clumps = cl; r.ops = op; r.n_clumps = len(recs); r.n_ops = len(ops)
    return r, (cs, cl, op)


# The device filter against the host's on synthetic clump lists (ygpu_inject_results): ties by the hundred, duplicates, copies, nests, up to 1 792 clumps a read (the
# largest the device stage sorts in its 64 KB of LDS) and beyond (the hand-over path), under the filter's parameter sets.  What is compared is the SAM text of both.
@pytest.mark.parametrize("seed,max_clumps,extra", [(1, 60, []), (2, 300, []), (3, 460, ["-FBS", "Y", "-PSS", "0.5", "-PRL", "0.5"]), (4, 120, ["-BP", "11", "-MGDP", "7", "-MNO", "5"]),
                                                   (5, 200, ["-FBS", "Y", "-MNO", "60", "-GOC", "3", "-GEC", "1", "-RC", "2"]), (6, 40, ["-oss", "stdout", "-FBS", "Y", "-PSS", "0.1", "-PRL", "0.1"]),
                                                   (7, 448, []), (8, 448, ["-FBS", "Y"]), (9, 224, ["-BP", "1", "-MGDP", "2"]), (10, 112, ["-MNO", "1", "-FBS", "Y", "-PSS", "0.99", "-PRL", "0.99"]),
                                                   (11, 30, ["-BP", "40", "-MGDP", "9", "-MS", "3", "-RC", "7"]), (12, 330, ["-MNO", "120"]),
                                                   (13, 1500, []), (14, 1792, ["-FBS", "Y"]), (15, 2100, ["-FBS", "Y", "-PSS", "0.5", "-PRL", "0.5"])])
def test_postfilter_on_synthetic_clump_lists(work, index11, seed, max_clumps, extra, monkeypatch):
    """Device stage vs the host's copy of the same routine (oqc_core.h) on clump lists no real read produces; as above, the pin to the reference is the host
    copy's (goldens, live reference) and the command line's with the device filter -- this test only shows that the two compilations agree where real reads do not go.
    Every third set with the reads' nodes and tables in HBM (YGPU_OQC_HBM=1): the instantiation of a read whose survivors do not fit its LDS."""
    import numpy as np
    rng = np.random.default_rng(seed)
    if seed % 3 == 0:
        monkeypatch.setenv("YGPU_OQC_HBM", "1")
    with ya.Session(["-x", index11, "-q", os.path.join(work, "r1k.fa"), "-osh", "stdout"] + list(extra)) as s:
        with ya.Context(s.index, s.params) as ctx:
            ctx.set_postfilter(s)
            b = s.next_batch(48)
            ctx.upload(b)
            r, _keep = _synthetic_results(s, b, rng, max_clumps)
            ctx.inject_results(r)
            host = s.emit(r)
            f = ctx.postfilter()
            assert f.n_reads == b.n_reads
            dev = s.emit_filtered(f)
            assert dev == host
            assert len(host) > 1000


@pytest.mark.parametrize("reads,extra", [("rchim.fa", []), ("r1k.fa", ["-FBS", "Y"])])
def test_postfilter_of_a_snapshot_while_the_next_batch_runs(work, index11, reads, extra):
    """The stage's two-thread use (include/yaha_hip.h: ygpu_postfilter_snapshot on the context's thread, ygpu_postfilter + ygpu_collect_filtered on another while the
    context uploads and runs the next batch -- here a DIFFERENT batch, so that a stage that read the context's live buffers instead of its snapshot would show):
    the filtered batch == the host's filter over the same results, five rounds."""
    import threading
    with ya.Session(["-x", index11, "-q", os.path.join(work, reads), "-osh", "stdout"] + list(extra)) as s, ya.Session(["-x", index11, "-q", os.path.join(work, "r10k.fa")]) as other:
        with ya.Context(s.index, s.params) as ctx:
            ctx.set_postfilter(s)
            b, nxt = s.next_batch(220), other.next_batch(40)
            ctx.upload(b); ctx.run()
            host = s.emit(ctx.collect())
            assert len(host) > 200
            for _ in range(5):
                ctx.upload(b); ctx.run(); ctx.postfilter_snapshot()
                got = {}
                t = threading.Thread(target=lambda: got.setdefault("f", ctx.postfilter()))
                t.start()
                ctx.upload(nxt); ctx.run()                 # the context's buffers now hold another batch's reads and results
                t.join()
                assert "f" in got and got["f"].n_reads == b.n_reads
                assert s.emit_filtered(got["f"]) == host
            with pytest.raises(RuntimeError):               # one snapshot at a time
                ctx.postfilter_snapshot(); ctx.postfilter_snapshot()
            ctx.postfilter()


def test_real_human_sequence_command_line_equals_the_reference(tmp_path):
    """The whole command line (HIP hot path, post-filter on the device) on real human sequence == SAM of the real reference (tests/golden/human: a 10 Mbp
    pseudo-reference of hg18 reads from the reference's testdata, 2 000 real 1 kbp + 2 000 real 200 bp hg18 reads that reach it through its repeat families, 1 500
    reads sampled from it), -L 15, defaults; line for line, in the reference's -t 1 order.  Also with the filter on the host."""
    from test_oracle_golden import unpack_human, human_golden
    d = str(tmp_path)
    meta = unpack_human(d)
    ya.build_index(["-g", os.path.join(d, "hs_pseudo.fa"), "-L", "15"])
    idx = os.path.join(d, "hs_pseudo.X15_01_65525S")
    for name, run in sorted(meta["runs"].items()):
        for extra in ([], ["-dpf", "N"]):
            p = subprocess.run([ya.CLI_PATH, "-x", idx, "-q", os.path.join(d, run["reads"]), "-osh", "stdout"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert p.returncode == 0, p.stderr.decode()[-2000:]
            assert strip_pg(p.stdout.decode()) == human_golden(name), "command line differs from the reference on %s %s" % (name, extra)
    # the hot path's work counters on this set, next to the synthetic genome's
    with ya.Session(["-x", idx, "-q", os.path.join(d, "hs_real1k.fa")]) as s:
        b = s.next_batch(2000)
        with ya.Context(s.index, s.params) as ctx:
            ctx.upload(b); ctx.run(); r = ctx.collect()
            ro, _own = oracle.run(s.index, s.params, b, threads=8)
            assert ya.result_records(r) == ya.result_records(ro) and r.counters.as_dict() == ro.counters.as_dict()
            c = r.counters.as_dict(); n = float(b.n_reads)
            print("human 1 kbp reads on the 10 Mbp pseudo-reference, per read: hits %.0f, fragments %.0f, clumps aligned %.0f, X-drop calls %.0f, extension cells %.0f, gap cells %.0f" %
                  (c["hits"] / n, c["fragments"] / n, c["clumps_formed"] / n, c["dp_ext_calls"] / n, c["dp_ext_cells"] / n, c["dp_gap_cells"] / n))
    os.remove(idx)


@pytest.mark.parametrize("n,bits", [(1, 3), (7, 7), (8, 8), (8191, 8), (8192, 12), (8193, 10), (12287, 9), (12288, 9), (12289, 9), (24575, 8), (24576, 12), (24577, 10), (100003, 16), (5000000, 7),
                                    (33000001, 10)])
def test_own_sums_and_orderings(work, index11, n, bits):
    """device/scan.h -- the exclusive sums (single pass, decoupled look-back, u32 and u64, in place) and the orderings by a small key that lay out the batch's
    variable-size outputs -- against plain host loops: tile edges (a tile is 512 threads x 48 u32 = 24 576 elements, 12 288 of u64; an ordering's 8 192), one element, tens of millions (1 343 tiles), sums beyond 2^32, keys with an
    offset and a shift, crowded buckets; every call twice over the same work words (they clean themselves up).  The same call runs A2's workgroup sort (device/wgsort.h) in
    five shapes and both rankings -- LDS atomics and ballots -- over segments of every fill and of diagonals made to tie (one value, two alternating lane by lane, equal
    inside a row, crowded neighbourhoods) against std::stable_sort, and wants the sort's own order check silent."""
    with ya.Session(["-x", index11, "-q", os.path.join(work, "r1k.fa")]) as s:
        with ya.Context(s.index, s.params) as c:
            for seed in (1, 2):
                c.selftest_primitives(n, seed=seed, key_bits=bits)
