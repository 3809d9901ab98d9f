"""GPU tier: more than one index image, more than one device.

The reference runs N threads over ONE mapped index (Query.c:565-626, 642-690).  Here N devices each need the image in HBM and take it through ygpu_init_multi: the
first from the host, every further one piece by piece from the device before it.  On the 1-GPU box the chain is driven with the same device listed twice (two
images on one device, the second one a device-to-device copy of the first); the tests marked `two_devices` run where the box has a second GPU -- `yaha -gpus 2`
against the real reference, a context on device 1, the peer copy itself and its host fall-back, and `bench.py --gpus 2` over RCCL."""
import json
import os
import subprocess
import sys

import pytest

import oracle
import yaha_amd as ya
from conftest import ROOT, strip_pg, golden_lines
from test_gpu_long_reads import by_read

pytestmark = pytest.mark.gpu
SIM = os.path.join(ROOT, "tools", "yaha_sim")


def _ndev():
    try:
        return ya.device_count()
    except Exception:
        return 0


two_devices = pytest.mark.skipif(_ndev() < 2, reason="needs a box with two GPUs")


def _golden_batch_equals_oracle(s, b, ctxs):
    exp, _own = oracle.run(s.index, s.params, b, threads=4)
    exp = ya.result_records(exp)
    for c in ctxs:
        c.upload(b)
        c.run()
        assert ya.result_records(c.collect()) == exp


def test_two_images_on_one_device_the_second_copied_from_the_first(work, index11):
    with ya.Session(["-x", index11, "-q", os.path.join(work, "rchim.fa")]) as s:
        b = s.next_batch(300)
        ctxs = ya.Context.on_devices(s.index, s.params, [0, 0], ctx_per_device=2)     # [image A: two contexts, image B (copied from A): two contexts]
        try:
            clone = ya.Context(s.index, s.params, parent=ctxs[3])          # and a context made later that shares the COPIED image
            _golden_batch_equals_oracle(s, b, ctxs + [clone])
            clone.close()
        finally:
            for c in reversed(ctxs):
                c.close()


def test_context_memory_accounting(work, index11):
    """ygpu_memory: the device's free and total bytes, and what a context's own buffers hold -- the shared index image counts for the context that owns it."""
    with ya.Session(["-x", index11, "-q", os.path.join(work, "r1k.fa")]) as s:
        b = s.next_batch(120)
        ctxs = ya.Context.on_devices(s.index, s.params, [0], ctx_per_device=2)
        try:
            f0, t0, own0 = ctxs[0].memory(); _f, _t, clone0 = ctxs[1].memory()
            image = s.index.n_base_bytes + 4 * s.index.totalMatches + 4 * (4 ** s.index.wordLen + 1)
            assert 0 < f0 < t0 and own0 >= image and clone0 < own0 - image + (1 << 20)
            _golden_batch_equals_oracle(s, b, ctxs)
            f1, _t1, own1 = ctxs[0].memory(); _f2, _t2, clone1 = ctxs[1].memory()
            assert own1 > own0 and clone1 > clone0 and f1 < f0                 # the arenas of a first batch
        finally:
            for c in reversed(ctxs):
                c.close()


def test_presized_and_parked_contexts(work, index11):
    """ygpu_get_arena_profile / ygpu_presize / ygpu_park (what the command line does with a device's further contexts): a context that takes the first one's arena
    capacities in one go holds as much as the first one before it has seen a read, computes the same results, and grows nothing on its first batch; a parked context
    gives its arenas back and refuses further batches."""
    with ya.Session(["-x", index11, "-q", os.path.join(work, "r10k.fa")]) as s:
        b = s.next_batch(12)
        ctxs = ya.Context.on_devices(s.index, s.params, [0], ctx_per_device=4)
        try:
            _golden_batch_equals_oracle(s, b, ctxs[:1])
            prof = ctxs[0].arena_profile()
            image = s.index.n_base_bytes + 4 * s.index.totalMatches + 4 * (4 ** s.index.wordLen + 1)
            _f, _t, own0 = ctxs[0].memory(); _f, _t, before = ctxs[1].memory()
            ctxs[1].presize(prof)
            _f, _t, after = ctxs[1].memory()
            assert prof.n > 100 and prof.bases > 0
            assert before < (own0 - image) // 4 and 0.9 * (own0 - image) <= after <= 1.1 * (own0 - image)
            _golden_batch_equals_oracle(s, b, ctxs[1:2])
            _f, _t, after_run = ctxs[1].memory()
            assert after_run == after                                         # nothing grew: the first batch ran in the presized arenas
            free0, _t, _o = ctxs[2].memory()
            _golden_batch_equals_oracle(s, b, ctxs[2:3])                       # a context that grows its own arenas, then is parked
            _f, _t, own2 = ctxs[2].memory()
            ctxs[2].park()
            free2, _t, parked = ctxs[2].memory()
            assert parked < own2 // 8 and free2 > free0 - (own2 // 8)
            with pytest.raises(RuntimeError, match="parked"):
                ctxs[2].upload(b)
            _golden_batch_equals_oracle(s, b, ctxs[:2])                        # the others are unaffected
            # a profile taken after the post-filter has run holds that stage's buffers too -- among them the look-back words of its exclusive sums, which their kernel
            # expects zeroed: a context presized from it filters its first batch like the first context does (round 5: it did not -- "a look-back gave up", or a fault)
            import ctypes as C
            def filtered(c):
                c.set_postfilter(s); c.upload(b); c.run(); f = c.postfilter()
                return int(f.n_clumps), int(f.n_ops), bytes(C.string_at(c._f_cs, 4 * (b.n_reads + 1))), bytes(C.string_at(c._f_cl, 40 * int(f.n_clumps))), bytes(C.string_at(c._f_ops, 4 * int(f.n_ops)))
            want = filtered(ctxs[0])
            ctxs[3].presize(ctxs[0].arena_profile())
            for _ in range(3):
                assert filtered(ctxs[3]) == want
        finally:
            for c in reversed(ctxs):
                c.close()


def test_three_images_chain_and_the_host_fall_back(work, index11, monkeypatch):
    with ya.Session(["-x", index11, "-q", os.path.join(work, "r1k.fa")]) as s:
        b = s.next_batch(120)
        for peer in ("1", "0"):                                              # YGPU_PEER_COPY=0: every image from the host
            monkeypatch.setenv("YGPU_PEER_COPY", peer)
            ctxs = ya.Context.on_devices(s.index, s.params, [0, 0, 0])
            try:
                _golden_batch_equals_oracle(s, b, ctxs[1:])
            finally:
                for c in reversed(ctxs):
                    c.close()


@pytest.mark.parametrize("fail_at", ["0", "1", "2", "1:2"])
def test_a_broken_peer_copy_falls_back_to_the_host(work, index11, monkeypatch, capfd, fail_at):
    """YGPU_PEER_FAIL_AT=k[:j]: piece k of the chain of device-to-device copies fails (on every chained device, or on the j-th device of the call only).  The device
    it fails on keeps the pieces it has, takes the others from the host, goes on serving the device behind it -- and every image is the reference's index: the
    batch's results equal the oracle's on all three.  (What a second GPU that refuses a hipMemcpyPeerAsync would look like; no second GPU is needed to test it.)"""
    monkeypatch.setenv("YGPU_PEER_FAIL_AT", fail_at)
    with ya.Session(["-x", index11, "-q", os.path.join(work, "r1k.fa")]) as s:
        b = s.next_batch(120)
        ctxs = ya.Context.on_devices(s.index, s.params, [0, 0, 0])
        try:
            err = capfd.readouterr().err
            assert "broke at piece %s" % fail_at.split(":")[0] in err and "the rest comes from the host" in err, err
            assert err.count("broke at piece") == (1 if ":" in fail_at else 2), err
            _golden_batch_equals_oracle(s, b, ctxs)
        finally:
            for c in reversed(ctxs):
                c.close()


def test_init_multi_reports_the_device_that_does_not_exist(work, index11):
    with ya.Session(["-x", index11, "-q", os.path.join(work, "r1k.fa")]) as s:
        with pytest.raises(RuntimeError, match="device 63: -2 device index out of range"):
            ya.Context.on_devices(s.index, s.params, [0, 63])


def test_command_line_with_two_logical_devices_on_one_gpu(work, index11):
    """`yaha -gpus 2 -ctx 2` with YAHA_DEVICES=0,0: the command line's whole multi-device path (ygpu_init_multi with two images, batches dealt to the contexts of both,
    output restored to input order) on the box's one GPU, against the reference's golden SAM."""
    p = subprocess.run([ya.CLI_PATH, "-x", index11, "-q", os.path.join(work, "rchim.fa"), "-osh", "stdout", "-gpus", "2", "-ctx", "2", "-batch", "25"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, YAHA_STATS="1", YAHA_DEVICES="0,0"))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert strip_pg(p.stdout.decode()) == golden_lines("rchim_default")
    st = [l for l in p.stderr.decode().split("\n") if l.startswith("[yaha] stats ")]
    stats = json.loads(st[0][len("[yaha] stats "):])
    assert stats["gpus"] == 2 and stats["ctx_per_gpu"] == 2 and len(stats["reads_per_device"]) == 2 and all(n > 0 for n in stats["reads_per_device"])


def test_eight_logical_devices_shake_the_host_path(tmp_path):
    """The host side of an 8-GPU node on the 1-GPU box: `YAHA_DEVICES=0,0,0,0,0,0,0,0 yaha -gpus 8 -ctx 1` -- eight index images (seven device-to-device copies behind one
    upload), eight context threads fed by one splitter / parser pool and drained by one formatter pool / ordered writer (the reference's parallel driver: Query.c:642-690)
    -- on a 40 Mbp -L 15 index, 131 072 reads of 1 kbp in batches of 2 048: every logical device takes reads, and the SAM is byte-identical to the one-device run."""
    g = str(tmp_path / "g.fa")
    subprocess.check_call([SIM, "genome", "--seed", "777", "--out", g, "--seqs", "8", "--len", "40000000", "--repeat-frac", "0.45"])
    ya.build_index(["-g", g, "-L", "15"])
    idx = str(tmp_path / "g.X15_01_65525S")
    reads = str(tmp_path / "r.fa")
    subprocess.check_call([SIM, "reads", "--genome", g, "--out", reads, "--seed", "92", "--n", "131072", "--len", "1000", "--div", "0.017", "--chimeric", "0.05"])
    one, eight = str(tmp_path / "one.sam"), str(tmp_path / "eight.sam")
    subprocess.check_call([ya.CLI_PATH, "-x", idx, "-q", reads, "-osh", one, "-batch", "2048"], stderr=subprocess.DEVNULL)
    p = subprocess.run([ya.CLI_PATH, "-x", idx, "-q", reads, "-osh", eight, "-gpus", "8", "-ctx", "1", "-batch", "2048"], stderr=subprocess.PIPE, env=dict(os.environ, YAHA_STATS="1", YAHA_DEVICES="0,0,0,0,0,0,0,0"))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    st = [l for l in p.stderr.decode().split("\n") if l.startswith("[yaha] stats ")]
    stats = json.loads(st[0][len("[yaha] stats "):])
    assert stats["gpus"] == 8 and stats["ctx_per_gpu"] == 1 and len(stats["reads_per_device"]) == 8 and all(n > 0 for n in stats["reads_per_device"]), stats
    assert sum(stats["reads_per_device"]) == stats["reads"] == 131072 and stats["ctx_left_out"] == 0
    a, b = strip_pg(open(one, newline="").read()), strip_pg(open(eight, newline="").read())
    assert len(a) > 131072 and a == b


@two_devices
def test_context_on_the_second_device(work, index11):
    with ya.Session(["-x", index11, "-q", os.path.join(work, "rchim.fa")]) as s:
        b = s.next_batch(300)
        with ya.Context(s.index, s.params, device=1) as c1:
            with ya.Context(s.index, s.params, parent=c1) as c1b:
                _golden_batch_equals_oracle(s, b, [c1, c1b])


@two_devices
def test_image_copied_between_two_devices_and_uploaded_to_both(work, index11, monkeypatch):
    with ya.Session(["-x", index11, "-q", os.path.join(work, "rchim.fa")]) as s:
        b = s.next_batch(300)
        for peer in ("1", "0"):
            monkeypatch.setenv("YGPU_PEER_COPY", peer)
            ctxs = ya.Context.on_devices(s.index, s.params, [0, 1])
            try:
                _golden_batch_equals_oracle(s, b, ctxs)
            finally:
                for c in reversed(ctxs):
                    c.close()


@two_devices
def test_command_line_over_two_devices_against_the_reference(work, tmp_path):
    """`yaha -gpus 2 -ctx 2` on 4 096 reads of 1 kbp at -L 15 (BASELINE config 4's shape, two shards) == the real reference (or the oracle port without it)."""
    g = str(tmp_path / "g.fa")
    subprocess.check_call([SIM, "genome", "--seed", "4242", "--out", g, "--seqs", "6", "--len", "20000000", "--repeat-frac", "0.4"])
    ya.build_index(["-g", g, "-L", "15"])
    idx = str(tmp_path / "g.X15_01_65525S")
    reads = str(tmp_path / "r.fa")
    subprocess.check_call([SIM, "reads", "--genome", g, "--out", reads, "--seed", "91", "--n", "4096", "--len", "1000", "--div", "0.017", "--chimeric", "0.1"])
    mine = str(tmp_path / "mine.sam")
    p = subprocess.run([ya.CLI_PATH, "-x", idx, "-q", reads, "-osh", mine, "-gpus", "2", "-ctx", "2", "-batch", "512"], stderr=subprocess.PIPE, env=dict(os.environ, YAHA_STATS="1"))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    st = [l for l in p.stderr.decode().split("\n") if l.startswith("[yaha] stats ")]
    stats = json.loads(st[0][len("[yaha] stats "):])
    assert stats["gpus"] == 2 and all(n > 0 for n in stats["reads_per_device"])          # both devices took batches
    ref = str(tmp_path / "ref.sam")
    if oracle.have_reference():
        oracle.run_reference(["-x", idx, "-q", reads, "-osh", ref, "-t", str(min(64, os.cpu_count() or 1))])
        exp = strip_pg(open(ref, newline="").read())
    else:
        exp = strip_pg(subprocess.run([ya.CLI_PATH, "-x", idx, "-q", reads, "-osh", "stdout"], stdout=subprocess.PIPE, check=True).stdout.decode())
    h1, r1, o1 = by_read(strip_pg(open(mine, newline="").read()))
    h2, r2, _ = by_read(exp)
    assert h1 == h2 and r1 == r2
    names = [l[1:].split()[0] for l in open(reads) if l.startswith(">")]
    assert o1 == [n for n in names if n in r1]                                            # input order over both devices


@two_devices
def test_bench_two_gpus_over_rccl(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "YAHA_BENCH_BACKEND")}
    env.update(YAHA_BENCH_CACHE=os.environ.get("YAHA_BENCH_CACHE", str(tmp_path / "cache")), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--genome-mbp", "100", "--contexts", "2", "--e2e-reads", "65536"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().split("\n") if l.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "weak" and j["value"] > 0
    e = j["end_to_end"]
    assert "error" not in e and e["gpus"] == 2 and len(e["reads_per_device"]) == 2 and all(n > 0 for n in e["reads_per_device"])


def test_bench_eight_ranks_on_one_gpu_over_gloo(tmp_path):
    """`python bench.py --gpus 8` as the driver starts it on an 8-GPU node -- eight ranks started by the bench itself, rendezvous on 127.0.0.1, barrier and
    max-over-ranks, rank 0's single JSON line -- on the ONE GPU of this box: YAHA_BENCH_BACKEND=gloo puts torch.distributed on the CPU and the ranks on device
    rank % device_count.  Nothing is stubbed: every rank builds its own context on the 100 Mbp index and runs the real hot path on its own shard of reads.  Not a
    measurement (eight processes share one device) -- what it proves is that the N = 8 launch path cannot fail for a trivial reason on first contact with real
    hardware: ranks_seen == 8, one line, reads of all ranks counted."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(YAHA_BENCH_CACHE=os.environ.get("YAHA_BENCH_CACHE", str(tmp_path / "cache")), HSA_ENABLE_IPC_MODE_LEGACY="0", YAHA_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--genome-mbp", "100", "--contexts", "1", "--reads-per-gpu", "2048",
                        "--blocks", "1", "--no-extras", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().split("\n") if l.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["ranks_seen"] == 8 and j["steps"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["reads_per_gpu"] == 2048 and abs(j["value"] * j["ms_per_step"] * 1e-3 - 8 * 2048) < 1.0      # whole-job rate: the reads of all eight ranks over the slowest rank's time
