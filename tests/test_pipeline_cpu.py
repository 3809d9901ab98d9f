"""CPU tier: the command line's host pipeline as a whole (host/pipeline.cpp: splitter -> parsers -> context threads -> formatter pool -> ordered writer, recycled
batch objects, -gpus N x -ctx M, failure handling), with the device entry points answered by a TEST DOUBLE backed by the oracle (tests/fixtures/oracle_device.cpp).
The program is built here into a scratch directory -- once under ThreadSanitizer, once under AddressSanitizer + UBSan -- and its SAM is compared with the
reference's golden output.  The HIP path itself is proven by the -m gpu tests; this file proves the host threads around it on a box without a GPU."""
import glob
import os
import subprocess

import pytest

from conftest import ROOT, golden_lines, strip_pg

HOST = os.path.join(ROOT, "yaha_amd", "csrc", "host")
SRCS = sorted(glob.glob(os.path.join(HOST, "*.cpp"))) + [os.path.join(ROOT, "yaha_amd", "csrc", "main.cpp"), os.path.join(ROOT, "tests", "fixtures", "oracle_device.cpp"),
                                                          os.path.join(ROOT, "oracle", "hotpath.cpp")]


def _build(tmp, san):
    exe = os.path.join(tmp, "yaha_" + san.replace(",", "_"))
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-pthread", "-o", exe] + SRCS)
    return exe


@pytest.fixture(scope="module")
def exes(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("pipe"))
    return {"tsan": _build(d, "thread"), "asan": _build(d, "address,undefined")}


def _run(exe, args, env=None, stdin=None):
    e = dict(os.environ, YAHA_KEEP_TEARDOWN="1", TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    e.update(env or {})
    return subprocess.run([exe] + args, env=e, stdin=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def _clean(p):
    err = p.stderr.decode()
    assert "ThreadSanitizer" not in err and "AddressSanitizer" not in err and "runtime error:" not in err, err[-4000:]


@pytest.mark.parametrize("san", ["tsan", "asan"])
@pytest.mark.parametrize("name,reads,extra", [
    ("rchim_default", "rchim.fa", ["-batch", "37", "-gpus", "2", "-ctx", "2"]),            # many small batches over 4 contexts on 2 devices: ordering, pool recycling
    ("r1k_default", "r1k.fa", []),                                                           # defaults: one batch of ~16 M bases, -ctx 3
    ("rq_default", "rq.fq", ["-batch", "5", "-t", "3"]),                                     # FASTQ, explicit formatter count
    ("r10k_default", "r10k.fa", ["-batch", "3", "-ctx", "1"]),
    ("rchim_default", "rchim.fa", ["-batch", "29", "-dpf", "N"]),                             # the host's post-filter by option
])
def test_command_line_pipeline_matches_the_reference(exes, work, index11, san, name, reads, extra):
    p = _run(exes[san], ["-x", index11, "-q", os.path.join(work, reads), "-osh", "stdout"] + extra, env={"YTEST_DEVICES": "2", "YAHA_CPUS": "6"})
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    _clean(p)
    assert strip_pg(p.stdout.decode()) == golden_lines(name)


def test_every_golden_run_through_the_command_line(exes, work, index11, meta):
    """All of the reference's golden outputs (defaults, -FBS Y, -OQC N, -o8, soft clipping, custom scoring, FASTQ ...) through the whole pipeline; with OQC on the
    post-filter stage (ygpu_postfilter: oqc_core.h behind the hot path) delivers what is printed, with -OQC N the host's duplicate removal does; and once more
    with the host filter forced (YAHA_HOST_OQC=1): the same text."""
    for name, r in sorted(meta["runs"].items()):
        args = ["-x", index11, "-q", os.path.join(work, r["reads"]), r["oflag"], "stdout", "-batch", "61"] + list(r["extra"])
        for env in ({}, {"YAHA_HOST_OQC": "1"}, {"YTEST_RAW_ABOVE": "3"}):        # the last one: reads of more than three clumps come back unfiltered and marked, as the device stage hands over its heaviest reads
            p = _run(exes["asan"], args, env=env)
            assert p.returncode == 0, (name, p.stderr.decode()[-1500:])
            _clean(p)
            assert strip_pg(p.stdout.decode()) == golden_lines(name), (name, env)


def test_stats_line_and_stdin(exes, work, index11):
    with open(os.path.join(work, "rchim.fa"), "rb") as f:
        p = _run(exes["tsan"], ["-x", index11, "-q", "stdin", "-osh", "stdout", "-batch", "50"], env={"YAHA_STATS": "1", "YAHA_READ_BLOCK": "997"}, stdin=f)
    assert p.returncode == 0
    _clean(p)
    assert strip_pg(p.stdout.decode()) == golden_lines("rchim_default")
    import json
    line = [l for l in p.stderr.decode().split("\n") if l.startswith("[yaha] stats ")]
    assert len(line) == 1
    st = json.loads(line[0][len("[yaha] stats "):])
    assert st["reads"] > 0 and st["last_batch_written_ms"] >= st["first_batch_written_ms"] > 0 and st["ctx_per_gpu"] == 3


def test_a_device_failure_stops_the_run_cleanly(exes, work, index11):
    # the third hot-path call fails: exit code 1, a message, and the output is a prefix of the good output that ends at a batch boundary
    good = _run(exes["tsan"], ["-x", index11, "-q", os.path.join(work, "rchim.fa"), "-osh", "stdout", "-batch", "20", "-ctx", "2"])
    bad = _run(exes["tsan"], ["-x", index11, "-q", os.path.join(work, "rchim.fa"), "-osh", "stdout", "-batch", "20", "-ctx", "2"], env={"YTEST_FAIL_RUN": "3"})
    _clean(bad)
    assert good.returncode == 0 and bad.returncode == 1
    assert "hot path failed" in bad.stderr.decode() and "stopping" in bad.stderr.decode()
    g, b = good.stdout.decode(), bad.stdout.decode()
    assert len(b) < len(g) and g.startswith(b) and (b.endswith("\n") or b == "")


def test_results_in_plain_memory_when_nothing_can_be_page_locked_and_the_fast_exit(exes, work, index11):
    p = _run(exes["asan"], ["-x", index11, "-q", os.path.join(work, "rchim.fa"), "-osh", "stdout", "-batch", "41"], env={"YTEST_NO_PINNED": "1", "YAHA_KEEP_TEARDOWN": ""})
    assert p.returncode == 0
    _clean(p)
    assert strip_pg(p.stdout.decode()) == golden_lines("rchim_default")


def test_contexts_without_room_for_their_arenas_are_left_out(exes, work, index11):
    # the device's first context holds 60 GB after its first batch, 50 GB are free: the other two contexts of -ctx 3 take no batches, the output is unchanged
    p = _run(exes["tsan"], ["-x", index11, "-q", os.path.join(work, "rchim.fa"), "-osh", "stdout", "-batch", "20", "-ctx", "3"], env={"YTEST_FREE_GB": "50", "YTEST_CTX_GB": "60", "YAHA_STATS": "1"})
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    _clean(p)
    err = p.stderr.decode()
    assert err.count("left out") == 2 and '"ctx_left_out": 2' in err
    assert strip_pg(p.stdout.decode()) == golden_lines("rchim_default")


@pytest.mark.parametrize("batch", ["100", "200", "500"])
def test_left_out_contexts_never_hold_a_batch(exes, work, index11, batch):
    """As many batches as contexts or fewer, and a ygpu_memory call that takes its time (a real hipSetDevice + hipMemGetInfo does): round 4's pipeline let a context
    take a batch, found no room, and pushed the batch back into a queue whose other consumers had already seen it empty and left -- a truncated SAM with exit
    code 0.  A context now decides before it takes a batch, and the run fails loudly if a batch is ever lost between two stages."""
    for _ in range(3):
        p = _run(exes["tsan"], ["-x", index11, "-q", os.path.join(work, "rchim.fa"), "-osh", "stdout", "-batch", batch, "-ctx", "3"],
                 env={"YTEST_FREE_GB": "50", "YTEST_CTX_GB": "60", "YTEST_MEMORY_MS": "5", "YAHA_STATS": "1"})
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        _clean(p)
        assert strip_pg(p.stdout.decode()) == golden_lines("rchim_default")
        assert '"ctx_left_out": 2' in p.stderr.decode()


@pytest.mark.parametrize("san", ["tsan", "asan"])
def test_the_filter_thread_overlaps_the_next_batch(exes, work, index11, san):
    """A context's post-filter runs on a thread of its own, on the snapshot the context's thread took, while that thread uploads and runs the next batch
    (pipeline.cpp: FilterSide).  A stage that takes its time (20 ms a batch in the double) and a snapshot that does: same SAM, in order, with one and with several
    contexts, and with the sequential order forced; a post-filter that fails stops the run as a failed hot path does."""
    base = ["-x", index11, "-q", os.path.join(work, "rchim.fa"), "-osh", "stdout", "-batch", "23"]
    for extra, env in ((["-ctx", "1"], {"YTEST_POSTFILTER_MS": "20"}), (["-ctx", "3", "-gpus", "2"], {"YTEST_POSTFILTER_MS": "7", "YTEST_SNAPSHOT_MS": "3", "YTEST_DEVICES": "2"}),
                       (["-ctx", "2"], {"YAHA_SERIAL_FILTER": "1", "YTEST_POSTFILTER_MS": "5"}), (["-ctx", "2", "-FBS", "Y"], {"YTEST_RAW_ABOVE": "3"})):
        args = base + extra
        if "-FBS" in extra:
            args[args.index("-osh")] = "-oss"                # (the golden with -FBS Y is a soft-clipped one)
        p = _run(exes[san], args, env=env)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        _clean(p)
        assert strip_pg(p.stdout.decode()) == golden_lines("rchim_FBS" if "-FBS" in extra else "rchim_default"), (extra, env)
    good = _run(exes[san], base + ["-ctx", "2"])
    bad = _run(exes[san], base + ["-ctx", "2"], env={"YTEST_FAIL_POSTFILTER": "4", "YTEST_POSTFILTER_MS": "5"})
    _clean(bad)
    assert good.returncode == 0 and bad.returncode == 1 and "injected post-filter failure" in bad.stderr.decode()
    g, b = good.stdout.decode(), bad.stdout.decode()
    assert len(b) < len(g) and g.startswith(b) and (b.endswith("\n") or b == "")


def test_no_such_device(exes, work, index11):
    p = _run(exes["asan"], ["-x", index11, "-q", os.path.join(work, "r1k.fa"), "-osh", "stdout", "-gpus", "2"], env={"YTEST_DEVICES": "1"})
    _clean(p)
    assert p.returncode == 1 and "ygpu_init(device 1) failed" in p.stderr.decode()
