#!/usr/bin/env python3
"""bench.py -- aligned reads/s of the MI355X hot path on BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (config.workload): hg18 is not available anywhere, so the run uses this repo's seeded
hg18-like synthetic genome (tools/yaha_sim.cpp: 41 % GC, 45 % of bases in diverged Alu-/L1-like repeat families,
calibrated to ~12 k seed hits and ~1.7 M X-drop cells per 1 kbp read), indexed with the
reference's defaults (-L 15 -S 1 -H 65525) by this repo's byte-identical indexer, and 1 000 bp reads with the
realised divergence of the bundled "E05" sets (1.7 %).  One step = one pass of the whole hot path (A1..A10:
k-mer lookup, seed join, chain DP, banded affine-gap DP + X-drop extension, score/split) over one batch of
reads that is already resident in HBM; results stay in HBM.  Every rank drives --contexts (default 4) device contexts on its
GPU, one host thread each, which take the K timed steps from a common counter: that many batches are in flight per GPU, so that one
context's latency-bound stages overlap the other's compute (contexts share nothing but the read-only index image).  Reads shard across ranks (weak scaling, fixed
reads per GPU), the index is replicated per GPU, there is no data-path collective: torch.distributed is used
for the barrier and the max-over-ranks only.

Adds to the JSON line:  "roofline" for the dominant kernel (k_ext_rows: the X-drop extension rows) from HIP events
on the stream it is launched on, and "cpu_baseline": the real reference binary (oracle/_ref/yaha -t <cores>) -- or,
where that is absent, the oracle port -- timed on a bounded sample of the same reads on this box's host cores.
"""
import argparse
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def owner_of(ticket, world):
    """Batches (tickets) are dealt round-robin to ranks / devices; reads are independent units."""
    return ticket % world


def merge_by_ticket(per_rank_parts):
    """Ordered merge of the ranks' outputs: identical to the single-rank (`-t 1`) order."""
    allp = {}
    for d in per_rank_parts:
        allp.update(d)
    return "".join(allp[k] for k in sorted(allp))


def max_over_ranks(value, dist, device="cuda"):
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def ensure_inputs(cache, genome_mbp, seed):
    """Genome FASTA + .nib2 + index in the cache directory (built once per node, reused by every rank/run)."""
    import yaha_amd as ya
    os.makedirs(cache, exist_ok=True)
    sim = os.path.join(ROOT, "tools", "yaha_sim")
    if not os.path.exists(sim):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", sim, os.path.join(ROOT, "tools", "yaha_sim.cpp")])
    tag = "g%dm_s%d" % (genome_mbp, seed)
    fa = os.path.join(cache, tag + ".fa")
    idx = os.path.join(cache, tag + ".X15_01_65525S")
    done = idx + ".done"
    if not os.path.exists(done):
        t = time.time()
        subprocess.check_call([sim, "genome", "--seed", str(seed), "--out", fa, "--seqs", "24", "--len", str(genome_mbp * 1000000), "--repeat-frac", "0.45", "--nrun", "6", "--lowcomplex", "8", "--repeat-bed", fa + ".repeats.bed"])
        ya.build_index(["-g", fa, "-L", "15"])
        open(done, "w").write("ok")
        log("built genome + index in %.1fs" % (time.time() - t))
    return fa, idx


def pick_genome_mbp(cache, seed):
    """G-hg18scale (3.1 Gbp, 4.3 GB table + 12 GB of offsets; genome + index build in ~30 s on the GPU, 21 GB of files) unless the box cannot hold it."""
    if os.path.exists(os.path.join(cache, "g3100m_s%d.X15_01_65525S.done" % seed)):
        return 3100
    try:
        os.makedirs(cache, exist_ok=True)
        free_disk = shutil.disk_usage(cache).free
        avail = 0
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
        if free_disk >= (36 << 30) and avail >= (96 << 30):
            return 3100
        log("G-hg18scale needs 36 GB of disk and 96 GB of memory (free: %.0f GB, %.0f GB): falling back to the 100 Mbp genome" % (free_disk / 2**30, avail / 2**30))
    except OSError as e:
        log("cannot size the box (%s): 100 Mbp genome" % e)
    return 100


def make_reads(cache, fa, tag, n, length, div, seed):
    path = os.path.join(cache, "%s_n%d_l%d_s%d.fa" % (tag, n, length, seed))
    if not os.path.exists(path):
        tmp = path + ".tmp%d" % os.getpid()
        subprocess.check_call([os.path.join(ROOT, "tools", "yaha_sim"), "reads", "--genome", fa, "--out", tmp, "--seed", str(seed), "--n", str(n), "--len", str(length), "--div", str(div)])
        os.replace(tmp, path)
    return path


def make_sv_reads(cache, fa, tag, seed, per=5, cov=1.5, n_ins=40):
    """BASELINE config 5's split-read set on the bench genome (tools/yaha_sim.cpp `sv`): RandomSV_Events.sim's four lines as they stand (DEL / DUP / INR / INV, 100 .. 10 000 bp
    in steps of 100) and INS lines for the genome's least diverged Alu-like copies (Alu_Insertions.sim's format), 500-mers at 2 %."""
    path = os.path.join(cache, "%s_sv_s%d_p%d.fa" % (tag, seed, per))
    if not os.path.exists(path):
        ev = path + ".events.sim"
        with open(ev, "w") as f:
            bed = fa + ".repeats.bed"
            if os.path.exists(bed):                      # (INS lines first: a bench batch is the file's first reads)
                rows = []
                for l in open(bed):
                    r = l.rstrip("\n").split("\t")
                    if len(r) == 6 and int(r[2]) - int(r[1]) >= 295 and float(r[5]) < 0.04:
                        rows.append(r)
                        if len(rows) >= 50 * n_ins:
                            break
                for r in sorted(rows, key=lambda r: float(r[5]))[:n_ins]:
                    f.write("INS\t%s\t%s\t%s\t%s\t%s\n" % (r[0], r[1], r[2], r[3], r[4]))
            f.write("DEL\t100\t10000\t100\nDUP\t100\t10000\t100\nINR\t100\t10000\t100\nINV\t100\t10000\t100\n")
        tmp = path + ".tmp%d" % os.getpid()
        subprocess.check_call([os.path.join(ROOT, "tools", "yaha_sim"), "sv", "--genome", fa, "--events", ev, "--out", tmp, "--seed", str(seed), "--per", str(per), "--cov", str(cov), "--div", "0.02"], stderr=subprocess.DEVNULL)
        os.replace(tmp, path)
    return path


def cli_run(ya, args, env_extra=None):
    """One run of the `yaha` command line; returns (seconds, stats dict of YAHA_STATS=1)."""
    env = dict(os.environ, YAHA_STATS="1"); env.update(env_extra or {})
    t = time.time()
    p = subprocess.run([ya.CLI_PATH] + list(args), stderr=subprocess.PIPE, check=True, env=env)
    dt = time.time() - t
    st = [l for l in p.stderr.decode().split("\n") if l.startswith("[yaha] stats ")]
    return dt, (json.loads(st[0][len("[yaha] stats "):]) if st else {})


def config5_leg(ya, idx, fa, cache, device, contexts, tag):
    """BASELINE config 5 on one GPU: the hot path on a resident batch of 32 768 SV 500-mers, the whole command line with -OQC Y -FBS Y on the whole set, and -- where the
    reference binary is present -- the command line's records for the set's first 8 192 reads against the reference's."""
    import oracle
    sv = make_sv_reads(cache, fa, tag, 5000, per=40)         # ~300 k reads of 500 bases: nine batches of 16 M bases for the command line (the reference's own sets hold 25 events a size)
    n_all = sum(1 for l in open(sv) if l.startswith(">"))
    out = side_workload(ya, idx, sv, 32768, device, contexts, 4, "c5: SV / repeat-insertion 500-mers at 2 % (RandomSV_Events.sim + Alu_Insertions.sim shapes), hot path", blocks=3)
    sam = "/dev/shm/yaha_bench_c5_%d.sam" % os.getpid()
    try:
        time.sleep(10)
        dt, st = cli_run(ya, ["-x", idx, "-q", sv, "-osh", sam, "-OQC", "Y", "-FBS", "Y"])
        run_ms = max(1.0, (st.get("total_ms") or 1e3 * dt) - (st.get("contexts_up_ms") or 0.0))
        out["command_line"] = {"reads": n_all, "seconds": dt, "e2e_reads_per_s": n_all / dt, "reads_per_s_after_contexts_up": n_all / (run_ms * 1e-3), "contexts_up_ms": st.get("contexts_up_ms"), "options": "-OQC Y -FBS Y",
                               "note": "nine batches: too few for the command line's own steady figure (reads after the first batch / time between the first and the last write)"}
        if oracle.have_reference():
            head, ref = os.path.join(cache, "c5_head.fa"), os.path.join(cache, "c5_head_reference.sam")
            head_reads(sv, head, 8192)
            oracle.run_reference(["-x", idx, "-q", head, "-osh", ref, "-OQC", "Y", "-FBS", "Y", "-t", str(usable_cpus())])
            names = set(l[1:].split()[0] for l in open(head) if l.startswith(">"))
            v = compare_sam(sam, ref, names); v["reads"] = 8192
            out["verified"] = v
            os.remove(head); os.remove(ref)
    finally:
        if os.path.exists(sam):
            os.remove(sam)
    return out


def host_ceiling(ya, idx, reads_path, n_devices=8):
    """What the host stages around the device sustain on this box, for the thread counts the command line would pick with `n_devices` devices (host/pipeline.cpp): the
    product's splitter + parsers (tools/host_ceiling.cpp) and its SAM formatter over device-filtered results of a real batch, replayed.  The smaller of the two is the
    rate up to which the host does not cap a node."""
    cpus = usable_cpus()
    n_parse = max(1, min(8, min(max((cpus + 7) // 10, (n_devices + 1) // 2), max(1, cpus // 3))))
    n_fmt = max(1, cpus - n_parse - 2)
    exe = os.path.join(ROOT, "tools", "host_ceiling")
    res = {"usable_cpus": cpus, "hardware_threads": os.cpu_count(), "devices_assumed": n_devices, "parser_threads": n_parse, "formatter_threads": n_fmt}
    lines = subprocess.run([exe, reads_path, "4096", str(n_parse)], stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
    for l in lines:
        if '"split+parse"' in l:
            res["parse_reads_per_s"] = json.loads(l)["reads_per_s"]
        elif '"split"' in l:
            res["split_reads_per_s"] = json.loads(l)["reads_per_s"]
    import threading
    N = 8192
    with ya.Session(["-x", idx, "-q", reads_path]) as s0:
        b0 = s0.next_batch(N)
        ctx = ya.Context(s0.index, s0.params)
        sessions = []
        try:
            ctx.upload(b0); ctx.run(); ctx.set_postfilter(s0); f = ctx.postfilter()
            sessions = [ya.Session(["-x", idx, "-q", reads_path, "-t", "1"]) for _ in range(n_fmt)]
            for x in sessions:
                assert x.next_batch(N).n_reads == N
            reps = 12

            def work(x):
                t_, n_ = C.c_char_p(), C.c_size_t()
                for _ in range(reps):
                    assert ya.lib().yaha_session_emit_filtered(x._h, C.byref(f), C.byref(t_), C.byref(n_)) == 0
            th = [threading.Thread(target=work, args=(x,)) for x in sessions]
            t = time.time()
            for x in th: x.start()
            for x in th: x.join()
            res["format_reads_per_s"] = N * n_fmt * reps / (time.time() - t)
        finally:
            for x in sessions:
                x.close()
            ctx.close()
    res["host_ceiling_reads_per_s"] = min(res.get("parse_reads_per_s", 0.0), res["format_reads_per_s"])
    return res


def eight_logical_devices(ya, cache, seed, n_reads=1048576):
    """The host path of an 8-GPU node on a 1-GPU box (Query.c:642-690 is the reference's parallel driver): `YAHA_DEVICES=0,0,0,0,0,0,0,0 yaha -gpus 8 -ctx 1` -- eight index
    images (a chain of seven device-to-device copies behind one upload), eight context threads, one splitter / parser pool / formatter pool / writer -- on the 100 Mbp index
    (eight images fit), 1 M reads.  The one physical GPU bounds the rate; what the leg shows is that the host side deals and re-orders for eight devices: every device gets
    reads, the output equals the one-device output."""
    fa, idx = ensure_inputs(cache, 100, seed)
    reads = make_reads(cache, fa, "g100m_8dev", n_reads, 1000, 0.017, 3200)
    one, eight = "/dev/shm/yaha_bench_1dev_%d.sam" % os.getpid(), "/dev/shm/yaha_bench_8dev_%d.sam" % os.getpid()
    try:
        time.sleep(15)
        dt1, st1 = cli_run(ya, ["-x", idx, "-q", reads, "-osh", one, "-batch", "4096"])
        time.sleep(15)
        dt8, st8 = cli_run(ya, ["-x", idx, "-q", reads, "-osh", eight, "-gpus", "8", "-ctx", "1", "-batch", "4096"], {"YAHA_DEVICES": "0,0,0,0,0,0,0,0"})
        same = subprocess.run(["cmp", "-s", one, eight]).returncode == 0
        if not same:                                      # (the @PG line echoes -gpus / -ctx when they are given: compare without it)
            same = [l for l in open(one) if not l.startswith("@PG")] == [l for l in open(eight) if not l.startswith("@PG")]
        return {"reads": n_reads, "index": "100 Mbp -L 15", "command": "YAHA_DEVICES=0,0,0,0,0,0,0,0 yaha -gpus 8 -ctx 1 -batch 4096", "seconds": dt8, "e2e_reads_per_s": n_reads / dt8, "steady_reads_per_s": st8.get("steady_reads_per_s"),
                "contexts_up_ms": st8.get("contexts_up_ms"), "reads_per_device": st8.get("reads_per_device"), "parsers": st8.get("parsers"), "formatters": st8.get("formatters"), "ctx_left_out": st8.get("ctx_left_out"),
                "one_device": {"command": "yaha -batch 4096 (-ctx 3)", "seconds": dt1, "steady_reads_per_s": st1.get("steady_reads_per_s")}, "output_identical_to_one_device": same}
    finally:
        for f in (one, eight):
            if os.path.exists(f):
                os.remove(f)


def head_reads(src, dst, n):
    with open(src) as f, open(dst, "w") as o:
        k = 0
        for line in f:
            if line.startswith(">"):
                k += 1
                if k > n:
                    break
            o.write(line)


def usable_cpus():
    """CPUs this process may really use: affinity mask and the control group's CPU quota (a 256-thread box with cpu.max = 16 CPUs runs 16 threads' worth)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(idx, fa, reads_path, n_reads, cache, target_s, read_len, div):
    """reference yaha on a bounded sample of the same workload (same simulator, genome, read length and divergence; its own seed so that it can be longer than one
    batch): best of -t {q, 2q, 4q}, q = the CPUs this box lets the process use (the reference serialises its readers on a file lock, and threads beyond the
    control group's quota only get throttled); at least target_s / 3 seconds of reference run time per thread count; wall time minus a one-read run (index
    mmap pre-touch)."""
    import oracle
    cores = os.cpu_count() or 1
    q = usable_cpus()
    if oracle.have_reference():
        one = os.path.join(cache, "one_read.fa")
        head_reads(reads_path, one, 1)
        t = time.time(); oracle.run_reference(["-x", idx, "-q", one, "-osh", "/dev/null", "-t", str(q)]); t_zero = time.time() - t
        probe = os.path.join(cache, "probe.fa")
        n_probe = min(n_reads, 64 * q)
        head_reads(reads_path, probe, n_probe)
        t = time.time(); oracle.run_reference(["-x", idx, "-q", probe, "-osh", "/dev/null", "-t", str(2 * q)]); t_probe = max(time.time() - t - t_zero, 1e-3)
        threads = sorted(set(min(t, cores) for t in (q, 2 * q, 4 * q)))
        # (the probe's rate is capped at 250 reads/s a usable CPU -- twice what the reference reaches on this workload: a probe that came out too fast, its one-read
        # run having paid for the index's page-in, once sized the sample for ten minutes of reference runs)
        n = int(min(262144, max(n_probe, (target_s / len(threads)) * min(n_probe / t_probe, 250.0 * q))))
        n = (n + 1023) // 1024 * 1024
        sample = make_reads(cache, fa, "cpu", n, read_len, div, 4000)
        by_t = {}
        ref_sam = os.path.join(cache, "cpu_sample_reference.sam")
        for nt in threads:
            # (the first thread count's SAM is kept -- a file in the page cache, ~2.3 KB a record -- for `verified` below; the others go to /dev/null as before)
            t = time.time(); oracle.run_reference(["-x", idx, "-q", sample, "-osh", ref_sam if nt == threads[0] else "/dev/null", "-t", str(nt)]); dt = max(time.time() - t - t_zero, 1e-3)
            by_t[nt] = n / dt
        best = max(by_t, key=lambda k: by_t[k])
        return {"value": by_t[best], "unit": "reads/s", "cores": best, "kind": "reference", "host_cores": cores, "usable_cpus": q, "reads_per_s_by_threads": {str(k): v for k, v in by_t.items()},
                "sample_path": sample, "sample_reads": n, "reference_sam": ref_sam,
                "sample": "%d reads of the same workload (%d bp, same simulator and genome, seed 4000), oracle/_ref/yaha (whole program: reader, hot path, OQC, SAM to /dev/null) at -t %s, best = -t %d; "
                          "wall minus a %.1fs one-read run; the box shows %d hardware threads and lets the process use %d CPUs (control-group quota)"
                          % (n, read_len, "/".join(str(t) for t in threads), best, t_zero, cores, q)}
    import yaha_amd as ya
    with ya.Session(["-x", idx, "-q", reads_path]) as s:
        b = s.next_batch(min(n_reads, 64 * cores))
        t = time.time(); oracle.run(s.index, s.params, b, threads=cores); dt = time.time() - t
        return {"value": b.n_reads / dt, "unit": "reads/s", "cores": cores, "kind": "port", "sample": "%d reads, oracle/hotpath.cpp on %d threads (hot path only)" % (b.n_reads, cores)}


def sam_by_read(path):
    """header lines without @PG, {QNAME: [records in order]}, QNAMEs in order of first appearance (the reference writes the reads of a -t N run in completion order)"""
    head, recs, order = [], {}, []
    with open(path, newline="") as f:
        for l in f:
            if l.startswith("@"):
                if not l.startswith("@PG"):
                    head.append(l)
                continue
            q = l.split("\t", 1)[0]
            if q not in recs:
                recs[q] = []; order.append(q)
            recs[q].append(l)
    return head, recs, order


def compare_sam(mine_path, ref_path, names=None):
    """Per-read comparison of two SAM files (records of a read in their order within the read; headers minus @PG).  names: restrict `mine` to these reads."""
    mh, mr, morder = sam_by_read(mine_path)
    rh, rr, _ = sam_by_read(ref_path)
    if names is not None:
        mr = {q: v for q, v in mr.items() if q in names}
    bad = [q for q in rr if mr.get(q) != rr[q]] + [q for q in mr if q not in rr]
    return {"reads_aligned": len(rr), "records": sum(len(v) for v in rr.values()), "records_mine": sum(len(v) for v in mr.values()), "headers_identical": mh == rh,
            "reads_differing": len(bad), "first_difference": (bad[0] if bad else None), "identical": (mh == rh and not bad and len(rr) > 0)}


def verify_against_reference(ya, idx, cpu, e2e_reads_path, cache, n_e2e=16384):
    """PARITY WHERE THE NUMBER IS MEASURED: the `yaha` command line of this repo (HIP hot path, device post-filter, defaults) on the very sample the reference was timed
    on -- same 3.1 Gbp index, >= 32 k reads -- compared per read with the SAM the reference wrote for it (Query.c:306-497 is the per-read loop both run); and the first
    n_e2e reads of the 1 M-read end-to-end input through the reference, compared with what the product wrote for those reads in a run over the whole file's head."""
    import oracle
    out = {}
    mine = os.path.join(cache, "cpu_sample_mine.sam")
    t = time.time()
    subprocess.run([ya.CLI_PATH, "-x", idx, "-q", cpu["sample_path"], "-osh", mine], stderr=subprocess.DEVNULL, check=True)
    out = compare_sam(mine, cpu["reference_sam"]); out["reads"] = cpu["sample_reads"]; out["seconds_product"] = time.time() - t
    out["what"] = "yaha (this repo, defaults, device post-filter) vs oracle/_ref/yaha on the cpu_baseline sample, %s index: per read, records in order; headers minus @PG" % os.path.basename(idx)
    os.remove(mine)
    if e2e_reads_path and os.path.exists(e2e_reads_path) and oracle.have_reference():
        head = os.path.join(cache, "e2e_head.fa"); head_reads(e2e_reads_path, head, n_e2e)
        ref2, mine2 = os.path.join(cache, "e2e_head_reference.sam"), os.path.join(cache, "e2e_head_mine.sam")
        oracle.run_reference(["-x", idx, "-q", head, "-osh", ref2, "-t", str(usable_cpus())])
        # (the product's records for the same reads, from a run over a head of the e2e file four batches long: the reads of interest share their batches with others)
        head4 = os.path.join(cache, "e2e_head4.fa"); head_reads(e2e_reads_path, head4, 4 * n_e2e)
        subprocess.run([ya.CLI_PATH, "-x", idx, "-q", head4, "-osh", mine2], stderr=subprocess.DEVNULL, check=True)
        names = set(l[1:].split()[0] for l in open(head) if l.startswith(">"))
        e = compare_sam(mine2, ref2, names); e["reads"] = n_e2e
        out["e2e_sample"] = e; out["identical"] = bool(out["identical"] and e["identical"])
        for f in (ref2, mine2, head, head4):
            os.remove(f)
    os.remove(cpu["reference_sam"])
    return out


def make_room(cache, need_gb=40.0):
    """The cache of the 3.1 Gbp workload takes 28 GB.  A box whose disk still holds the temporary directories of earlier (killed) test sessions of this user may not
    have that: they are scratch of this repo's own GPU tier (tests/, ~18 GB a session) and are removed when space is short.  (Round 5: a reused box ran full.)"""
    try:
        import getpass, glob, shutil, tempfile
        d = cache if os.path.isdir(cache) else (os.path.dirname(cache) or "/tmp")
        st = os.statvfs(d)
        if st.f_bavail * st.f_frsize / 1e9 >= need_gb or glob.glob(os.path.join(cache, "g3100m_*.X15_*")):
            return
        base = os.path.join(tempfile.gettempdir(), "pytest-of-%s" % getpass.getuser())
        # (not the session pytest-current points at, nor anything touched in the last half hour: a test session of this user may be running beside the bench)
        cur = os.path.realpath(os.path.join(base, "pytest-current")) if os.path.islink(os.path.join(base, "pytest-current")) else None
        for old in glob.glob(os.path.join(base, "pytest-*")):
            if os.path.islink(old) or os.path.realpath(old) == cur or time.time() - os.path.getmtime(old) < 1800:
                continue
            shutil.rmtree(old, ignore_errors=True)
    except Exception as e:                               # (never in the way of the measurement)
        sys.stderr.write("[bench] make_room: %s\n" % e)


def run_contexts(ctxs, steps, collect=False, postfilter=False):
    """`steps` passes of the hot path shared out to the contexts (one host thread each, a common counter); returns (seconds, summed stage ms)."""
    import threading
    stage_ms, lock, todo, flt, flt_err = {}, threading.Lock(), [steps], {}, []

    def stepper(c):
        while True:
            with lock:
                if todo[0] <= 0:
                    return
                todo[0] -= 1
            c.run()                                     # synchronous: returns when the results are complete in HBM
            if postfilter:
                # OQC / FBS / MAPQ on the device (oqc_stage.h) and D2H of the clumps that are printed -- on a thread of its own, on the stage's snapshot of the
                # results, while this thread runs the context's next step (as the command line does: host/pipeline.cpp FilterSide)
                if flt.get(c) is not None:
                    flt[c].join()
                    if flt_err:
                        raise flt_err[0]
                c.postfilter_snapshot()

                def filter_it(c=c):
                    try:
                        c.postfilter()
                    except Exception as e:              # (reported by the thread that joins)
                        flt_err.append(e)
                flt[c] = threading.Thread(target=filter_it); flt[c].start()
            elif collect:
                c.collect()                             # D2H of the clump records and edit ops into the context's host buffers
            tm = c.timing()[1]
            with lock:
                for k, v in tm.items():
                    stage_ms[k] = stage_ms.get(k, 0.0) + v
    t0 = time.time()
    th = [threading.Thread(target=stepper, args=(c,)) for c in ctxs]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for x in flt.values():
        x.join()
    if flt_err:
        raise flt_err[0]
    return time.time() - t0, stage_ms


def side_workload(ya, idx, reads_path, n_reads, device, contexts, steps, label, blocks=1, extra_args=()):
    """The same step on another read length (BASELINE configs 1 and 3), after the timed region: reads resident in HBM, results left in HBM."""
    with ya.Session(["-x", idx, "-q", reads_path] + list(extra_args)) as s:
        b = s.next_batch(n_reads)
        offs = C.cast(b.offsets, C.POINTER(C.c_uint64))
        n, bases = b.n_reads, int(offs[b.n_reads] - offs[0])
        ctxs = [ya.Context(s.index, s.params, device=device)]
        try:                                            # (whatever happens, the contexts are closed: a leg that fails must not leave ~200 GB of arenas behind for the legs after it)
            # every context takes its two warm-up passes (the first one sizes its arenas, the second one runs with them) before the next one is made: a context that
            # does not fit beside the others (larger batches than the headline's: 10 kbp reads, the c5 set) is left out instead of squeezing all of them
            ctxs[0].upload(b); ctxs[0].run(); ctxs[0].run()
            for k in range(1, max(1, contexts)):
                c = None
                try:
                    free_b, _tot, first = ctxs[0].memory()
                    image = int(s.index.n_base_bytes) + 4 * int(s.index.totalMatches) + 4 * (4 ** int(s.index.wordLen) + 1)      # (the first context's figure includes the index image)
                    if free_b < 1.1 * max(first - image, 1 << 30):
                        log("%s: %d contexts (%.0f GB free, a context's arenas hold %.0f GB)" % (label[:24], k, free_b / 1e9, (first - image) / 1e9))
                        break
                    c = ya.Context(s.index, s.params, device=device, parent=ctxs[0])
                    c.upload(b); c.run(); c.run()
                    ctxs.append(c)
                except Exception as e:
                    log("%s: context %d of %d left out (%s)" % (label[:24], k + 1, contexts, str(e)[:100]))
                    if c is not None:
                        c.close()
                    break
            dts, st = [], {}
            for _b in range(max(1, blocks)):
                d1, s1 = run_contexts(ctxs, steps); dts.append(d1)
                for k2, v in s1.items():
                    st[k2] = st.get(k2, 0.0) + v / max(1, blocks)
            dt = sorted(dts)[len(dts) // 2]
            cnt = ctxs[0].collect().counters.as_dict()
        finally:
            for c in reversed(ctxs):
                c.close()
    return {"workload": label, "contexts": len(ctxs), "reads_per_step": n, "steps": steps, "blocks": len(dts), "reads_per_s": n * steps / dt, "bases_per_s": bases * steps / dt, "ms_per_step": 1e3 * dt / steps, "ms_per_step_min": 1e3 * min(dts) / steps, "ms_per_step_max": 1e3 * max(dts) / steps,
            "k_ext_rows_ms_per_step": st.get("ext_rows_device_clock", 0.0) / steps, "dp_cells_per_read": (cnt["dp_ext_cells"] + cnt["dp_gap_cells"]) / max(n, 1), "hits_per_read": cnt["hits"] / max(n, 1)}


def end_to_end(ya, idx, fa, cache, n_reads, seed, gpus=1, read_len=1000, div=0.017, settle_s=20.0):
    """The whole `yaha` command line with its defaults (process start, index mmap + upload to every device, input parsing, device, OQC, SAM text to a file in
    /dev/shm) on n_reads x 1 kbp reads -- BASELINE config 4's size by default -- over `gpus` devices.  `steady_reads_per_s` is the command line's own figure
    (YAHA_STATS=1): reads written after the first batch / time between the first and the last batch's write, i.e. without start-up."""
    reads = make_reads(cache, fa, "e2e", n_reads, read_len, div, seed)
    if gpus > 1:                                         # N devices: N times the reads (BASELINE config 4's size per device), the one file repeated
        many = os.path.join(cache, "e2e_x%d_%s" % (gpus, os.path.basename(reads)))
        if not os.path.exists(many):
            with open(many + ".tmp", "wb") as o:
                for _ in range(gpus):
                    with open(reads, "rb") as f:
                        shutil.copyfileobj(f, o, 16 << 20)
            os.replace(many + ".tmp", many)
        reads, n_reads = many, n_reads * gpus
    out = "/dev/shm/yaha_bench_e2e_%d.sam" % os.getpid()
    tiny = os.path.join(cache, "tiny.fa")
    head_reads(reads, tiny, 16)
    more = ["-gpus", str(gpus)] if gpus > 1 else []
    env = dict(os.environ, YAHA_STATS="1")
    def settle():
        """The driver scrubs device memory a process has freed in the background, and a process that allocates meanwhile waits for it: a command line started right
        after the bench contexts (or the previous command line) freed ~200 GB spends seconds in its first allocations -- a property of what ran before it.  Tiny
        runs, 5 s apart, after 20 s of waiting, until the contexts are up within a second again (another 45 s at most)."""
        # (the arenas of a full run ask for ten times what the tiny run's index does: give the scrubbing its head start.  The scrubbing also runs ON the device: a command
        # line that starts while 240 GB of the previous one's arenas are still being cleared runs beside it -- the 10 kbp leg, whose arenas are the largest, took 2.6 s
        # instead of 1.6 s twenty seconds behind its predecessor, 3.0 s five seconds behind, 1.6 s after 28 s: tools/r06/cli_c3_after_load.sh -- so that leg waits 40 s)
        time.sleep(settle_s)
        t0 = time.time(); ups = []
        while True:
            p = subprocess.run([ya.CLI_PATH, "-x", idx, "-q", tiny, "-osh", out] + more, stderr=subprocess.PIPE, check=True, env=env)
            st = [l for l in p.stderr.decode().split("\n") if l.startswith("[yaha] stats ")]
            up = json.loads(st[0][len("[yaha] stats "):]).get("contexts_up_ms", 0.0) if st else 0.0
            ups.append(round(up))
            if up < 1000.0 or time.time() - t0 > 45: return ups
            time.sleep(5)
    try:
        runs, stats, settled = [], [], []
        for _ in range(2):                               # best of two, both reported
            settled.append(settle())
            t = time.time()
            p = subprocess.run([ya.CLI_PATH, "-x", idx, "-q", reads, "-osh", out] + more, stderr=subprocess.PIPE, check=True, env=env)
            runs.append(time.time() - t)
            st = [l for l in p.stderr.decode().split("\n") if l.startswith("[yaha] stats ")]
            stats.append(json.loads(st[0][len("[yaha] stats "):]) if st else {})
        best = min(range(len(runs)), key=lambda i: runs[i]); dt = runs[best]
        nrec = sum(1 for l in open(out) if not l.startswith("@"))
    finally:
        if os.path.exists(out):
            os.remove(out)
    per_dev = stats[best].get("reads_per_device") or []
    steady = stats[best].get("steady_reads_per_s")
    run_ms = max(1.0, (stats[best].get("total_ms") or 1e3 * dt) - (stats[best].get("contexts_up_ms") or 0.0))
    return {"reads": n_reads, "gpus": gpus, "seconds": dt, "e2e_reads_per_s": n_reads / dt, "steady_reads_per_s": steady, "reads_per_s_after_contexts_up": n_reads / (run_ms * 1e-3), "contexts_up_ms": stats[best].get("contexts_up_ms"), "reads_per_device": per_dev,
            "device_steady_reads_per_s": [round(steady * n / max(1, sum(per_dev))) for n in per_dev] if steady else None, "seconds_each_run": runs, "contexts_up_ms_of_the_settling_runs": settled, "sam_records": nrec, "cli_stats": stats[best],
            "read_len": read_len, "command": "yaha -x IDX -q %d_reads.fa -osh /dev/shm/out.sam%s (defaults: -ctx 3, batches of ~16 M bases, host threads from the usable CPUs)" % (n_reads, " -gpus %d" % gpus if gpus > 1 else "")}


def stub_rank(args, rank, world):
    """YAHA_BENCH_STUB=1 (tests only, CPU tier): the launch / rendezvous / barrier / max-over-ranks / one-JSON-line plumbing of the N > 1 path with a
    sleep in place of the device step -- no GPU, gloo instead of RCCL.  Never a measurement: the line says so."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    d = dist if world > 1 else None
    if d: d.barrier()
    t0 = time.time()
    for _ in range(args.steps):
        time.sleep(0.002 * (rank + 1))
    if d: d.barrier()
    dt = max_over_ranks(time.time() - t0, d, device="cpu")
    if rank == 0:
        print(json.dumps({"metric": "STUB (no device): launch-path test only", "value": world * args.reads_per_gpu * args.steps / dt, "unit": "reads/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "none",
                          "config": {"workload": "stub"}, "ranks_seen": world}), flush=True)
    if d:
        d.barrier(); d.destroy_process_group()


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as CHILD processes (torch.distributed.run, one rank per GPU,
    rendezvous on 127.0.0.1) before this process has touched a GPU -- it never does -- and relay rank 0's JSON line.  A process that has initialised the
    GPU is never replaced by another program (the pool forbids that exec); the parent stays a plain supervisor and exits with the children's code."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(argv)
    log("starting %d ranks: %s" % (n, " ".join(cmd)))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in p.stdout:                                   # rank 0 prints exactly one JSON line; anything else on stdout is passed on to stderr
        t = out.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        elif t:
            print(t, file=sys.stderr, flush=True)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    if rc != 0 or line is None:
        raise SystemExit(rc if rc != 0 else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads-per-gpu", type=int, default=16384)
    ap.add_argument("--read-len", type=int, default=1000)
    ap.add_argument("--genome-mbp", type=int, default=0, help="synthetic genome size; 0 = G-hg18scale (3 100 Mbp: the index the metric is quoted on) when the box has the disk and memory for it, else 100")
    ap.add_argument("--div", type=float, default=0.017)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--cpu-seconds", type=float, default=60.0, help="CPU-baseline budget (wall seconds of the reference over all its thread counts)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick-extras", action="store_true", help="of the legs after the timed region only the D2H- and post-filter-inclusive rates (for A/Bs of the filter stage)")
    ap.add_argument("--no-extras", action="store_true", help="skip the legs measured after the timed region (other read lengths, D2H-inclusive rate, command line)")
    ap.add_argument("--e2e-reads", type=int, default=1048576, help="reads of the end-to-end command-line leg (BASELINE config 4: 1 M x 1 kbp)")
    ap.add_argument("--contexts", type=int, default=4, help="device contexts (batches in flight) per GPU")
    ap.add_argument("--blocks", type=int, default=5, help="the timed region is this many back-to-back blocks of --steps steps; the median block is the headline")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])       # nothing above has imported torch or touched a GPU
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        log("--gpus %d but WORLD_SIZE=%d: the launcher's world size is what runs" % (args.gpus, world))
    if os.environ.get("YAHA_BENCH_STUB"):
        return stub_rank(args, rank, world)
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (YAHA_BENCH_BACKEND=gloo, with fewer GPUs than ranks: the ranks share devices and torch.distributed runs on the CPU -- a dry run of the N > 1 path on a 1-GPU box)
        backend = os.environ.get("YAHA_BENCH_BACKEND", "nccl")
        local = local % max(1, torch.cuda.device_count()) if backend == "gloo" else local
        torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        # the waits AFTER the timed region go through a CPU group: a rank parked in an RCCL barrier keeps a kernel spinning on its GPU, and rank 0 runs the whole
        # command line over all the devices meanwhile (end_to_end)
        cpu_group = dist.new_group(backend="gloo")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    import yaha_amd as ya
    cache = os.environ.get("YAHA_BENCH_CACHE", "/tmp/yaha_bench_cache")
    if rank == 0:
        make_room(cache)
        if args.genome_mbp <= 0:
            args.genome_mbp = pick_genome_mbp(cache, args.seed)
        fa, idx = ensure_inputs(cache, args.genome_mbp, args.seed)
    barrier()
    if args.genome_mbp <= 0:
        args.genome_mbp = pick_genome_mbp(cache, args.seed)              # rank 0 has built it by now: the same answer on every rank
    fa, idx = ensure_inputs(cache, args.genome_mbp, args.seed)
    reads_path = make_reads(cache, fa, "g%dm" % args.genome_mbp, args.reads_per_gpu, args.read_len, args.div, 1000 + rank)

    with ya.Session(["-x", idx, "-q", reads_path]) as s:
        b = s.next_batch(args.reads_per_gpu)
        n_reads = b.n_reads
        offs = C.cast(b.offsets, C.POINTER(C.c_uint64))
        n_bases = int(offs[n_reads] - offs[0])
        # args.contexts device contexts on this GPU (they share the index image), one host thread each; every step is one pass of
        # the whole hot path over the batch by ONE context, and the contexts take the K steps from a common counter: while one
        # is in a latency-bound stage the other one's kernels fill the device.
        # (Each context takes its first pass -- which allocates its arenas, ~58 GB at this batch -- before the next one is created: a context that does not
        # fit beside the others is left out, the line's `parallelism` says how many ran.)
        ctxs, t_up = [], 0.0
        for k in range(max(1, args.contexts)):
            c = None
            try:
                c = ya.Context(s.index, s.params, device=local) if k == 0 else ya.Context(s.index, s.params, device=local, parent=ctxs[0])
                t = time.time(); c.upload(b); t_up += time.time() - t
                c.run()
                ctxs.append(c)
            except Exception as e:
                if k == 0:
                    raise
                log("context %d of %d not created (%s): running with %d" % (k + 1, args.contexts, str(e)[:120], len(ctxs)))
                if c is not None:
                    c.close()
                break
        t_up /= len(ctxs)
        args.contexts = len(ctxs)
        for c in ctxs:
            for _ in range(args.warmup):
                c.run()
        # The timed region: args.blocks back-to-back blocks of EXACTLY args.steps steps, each bracketed by barrier + synchronize on both sides.  The headline is the
        # MEDIAN block (a 0.9 s block cannot resolve a 1 % kernel change against the box's run-to-run spread; five of them and their min / max can).
        block_dt, stage_ms = [], {}
        for _b in range(max(1, args.blocks)):
            barrier()
            t0 = time.time()
            _dt_inner, st_b = run_contexts(ctxs, args.steps)
            barrier()
            block_dt.append(time.time() - t0)
            for k2, v in st_b.items():
                stage_ms[k2] = stage_ms.get(k2, 0.0) + v / max(1, args.blocks)      # mean over the blocks of the per-block sums
        # ---- after the timed region ----
        t = time.time(); r = ctxs[0].collect(); t_down = time.time() - t
        try:
            tv = ctxs[0].trace_volume()          # the trace stream of the batch's last run: records written against records a traceback can visit
        except Exception as e:
            tv = None; log("trace volume: %s" % str(e)[:200])
        counters = r.counters.as_dict()
        n_clumps = int(r.n_clumps); n_ops = int(r.n_ops)
        d2h = None; unshared_rows_ms = None
        if world == 1 and not args.no_extras:
            # the dominant kernel with the chip to itself: two steps of one context (in the timed region the launches of the contexts share the CUs, and a launch's
            # duration -- first wave's start to last wave's end -- stretches by whatever the others' kernels take from it)
            _dt1, st1 = run_contexts(ctxs[:1], 2)
            unshared_rows_ms = st1.get("ext_rows_device_clock", 0.0) / 2 or None
            dt2, _st = run_contexts(ctxs, args.steps, collect=True)     # every step also copies its results to host memory (the other context computes meanwhile)
            d2h = {"value_with_d2h": n_reads * args.steps / dt2, "ms_per_step": 1e3 * dt2 / args.steps, "result_bytes_per_step": 32 * n_clumps + 4 * n_ops + 4 * (n_reads + 1)}
            # ... and with the post-filter on the device instead (what the command line does): every step = hot path + OQC/FBS/MAPQ of all reads + D2H of the printed clumps only
            try:
                for c in ctxs:
                    c.set_postfilter(s)
                f = ctxs[0].postfilter(); fb = 40 * int(f.n_clumps) + 4 * int(f.n_ops) + 4 * (n_reads + 1)
                dt3, _st3 = run_contexts(ctxs, args.steps, postfilter=True)
                d2h.update({"value_with_postfilter": n_reads * args.steps / dt3, "ms_per_step_with_postfilter": 1e3 * dt3 / args.steps, "filtered_result_bytes_per_step": fb, "printed_clumps_per_read": int(f.n_clumps) / n_reads})
            except Exception as e:
                d2h["postfilter_error"] = str(e)[:200]
        k = s.params.wordLen
        for c in reversed(ctxs):
            c.close()
    block_dt = [max_over_ranks(x, dist, device="cpu" if os.environ.get("YAHA_BENCH_BACKEND") == "gloo" else "cuda") for x in block_dt]     # a block lasts as long as its slowest rank
    dt = sorted(block_dt)[len(block_dt) // 2]
    ranks_seen = 1
    if dist is not None:
        one = torch.ones(1); dist.all_reduce(one, group=cpu_group); ranks_seen = int(one.item())      # every rank that reached the end of its timed region
        dist.barrier(group=cpu_group)                    # every rank has closed its contexts: the devices are free for the command-line leg below
    if rank != 0:
        if dist is not None:
            dist.barrier(group=cpu_group)
            dist.destroy_process_group()
        return

    steps = args.steps
    total_reads = world * n_reads * steps
    value = total_reads / dt
    # ALGORITHMIC bytes per read, SURVEY.md 8(d): read codes + 2 strands x 2 u32 table words per k-mer + hit words + touched reference nibbles
    # (lazy: rows computed + band per call + exact-match bases) + result records; H, R and the records are counted on the device for this input.
    Lq = n_bases / n_reads
    B = Lq + 16 * (Lq - k + 1) + 4 * counters["hits"] / n_reads + counters["ref_bases_touched"] / n_reads / 2 + (24 * counters["clumps_scored"] + 3 * counters["ops_out"]) / n_reads
    # Dominant kernel (profiles/): k_ext_rows, the X-drop extension rows of all reads of the batch, one problem per lane.  One launch processes the batch's
    # n_reads reads, so its algorithmic bytes are B x n_reads (SURVEY 8(d)'s per-read figure x the reads of one launch).  Duration: measured live in the
    # timed region -- HIP events around the launch on its stream (`ext_rows`) and the kernel's own wall_clock64() stamps (`ext_rows_device_clock`); with
    # several contexts per GPU the event bracket also contains the time the launch queues behind the other context's kernels, so the device-clock duration
    # (which is what rocprofv3 reports for the kernel, profiles/) is the one used.
    align_ms = stage_ms.get("align_dp", 0.0) / steps
    rows_ms = stage_ms.get("ext_rows", 0.0) / steps
    rows_dev_ms = stage_ms.get("ext_rows_device_clock", 0.0) / steps
    if rows_ms > 0:
        packed = stage_ms.get("ext_rows_packed16", 0.0) > 0       # a flag among the timings: the packed 16-bit kernel ran (16-byte trace records instead of 12-byte rows)
        kname, kernel_ms = "k_ext_rows", (rows_dev_ms if rows_dev_ms > 0 else rows_ms)
        stream_bytes = (17.0 if packed else 13.5) * counters["dp_ext_rows"] + 48.0 * counters["dp_ext_calls"]     # what the kernel itself streams: per row a query code (1/2 B packed in the 16-bit kernel, 1 B in the 32-bit one) + 1/2 B reference + 16 (12) B of trace cells; 48 B per problem
    else:                                                    # other band widths run the wave-per-root kernel
        packed = False
        kname, kernel_ms, stream_bytes = "k_align", align_ms, None
    kbytes = B * n_reads
    achieved = (kbytes / (kernel_ms * 1e-3)) / 1e9 if kernel_ms > 0 else 0.0
    ach_un = (kbytes / (unshared_rows_ms * 1e-3)) / 1e9 if (rows_ms > 0 and unshared_rows_ms) else None      # the same launch with the chip to itself
    # HBM traffic of that launch from the PMC counters: bench.py cannot read hardware counters of its own process, so this is the figure of the last
    # profiling pass of the same command (tools/pmc_pass.sh -> profiles/pmc_latest.json), used only when it was taken on the same batch size and kernel.
    traffic, pmc = None, None
    pj = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pj):
        try:
            pmc = json.load(open(pj))
            import hashlib
            dev = os.path.join(ROOT, "yaha_amd", "csrc", "device")
            ksrc = hashlib.sha256(b"".join(open(os.path.join(dev, f), "rb").read() for f in ("ext_lanes.h", "ext_lanes_pk.h"))).hexdigest()[:16]
            # counters of another kernel are worse than none: the file names the sources of the extension kernels it was measured on (tools/pmc_summary.py)
            if pmc.get("reads_per_gpu") == n_reads and pmc.get("kernel") == kname and pmc.get("kernel_source_sha16") == ksrc:
                traffic = pmc.get("hbm_bytes_per_launch")
            else:
                pmc = None
        except Exception:
            traffic, pmc = None, None
    valu = pmc.get("valu_insts_per_launch") if pmc else None
    simd_cycles = 1024 * 2.4e9 * (unshared_rows_ms if (rows_ms > 0 and unshared_rows_ms) else kernel_ms) * 1e-3   # 256 CUs x 4 SIMDs at 2.4 GHz over the launch (the counters are a one-context profile's: priced against the unshared duration when it was measured)
    out = {
        "metric": "aligned reads/s (whole node), 1 000 bp reads, OQC mode hot path", "value": value, "unit": "reads/s",
        "n_gpus": world, "ranks_seen": ranks_seen, "steps": steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / steps, "ms_per_step_min": 1e3 * min(block_dt) / steps, "ms_per_step_max": 1e3 * max(block_dt) / steps, "blocks": len(block_dt), "ms_per_step_blocks": [round(1e3 * x / steps, 3) for x in block_dt],
        "timing_note": "%d back-to-back blocks of exactly %d steps, each bracketed by barrier + synchronize, max over ranks per block; value and ms_per_step are the MEDIAN block's" % (len(block_dt), steps),
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": ("int16" if packed else "int32"), "dtype_note": "integer dynamic programming, bit-exact: the X-drop extension rows (93 % of the DP cells) in saturating packed int16 when the scores fit (else int32), everything else int32", "data": "synthetic",
        "bases_per_s": value * Lq,
        # (the driver's record keeps the first 120 characters of a string: read count, read length and the options the metric is quoted on come first)
        "config": {"workload": "%d x %d bp reads/GPU, -BW 5 -G 50, hg18-like %d Mbp index -L 15; BASELINE config 2 shape" % (n_reads, args.read_len, args.genome_mbp),
                   "workload_detail": "synthetic hg18-like genome %d Mbp (24 seqs, 45%% repeats), index -L 15 -S 1 -H 65525, %d x %d bp reads per GPU at %.1f%% divergence, defaults -BW 5 -G 50 -H 650 -M 25 -X 25, hot path A1..A10 with inputs resident in HBM"
                   % (args.genome_mbp, n_reads, args.read_len, 100 * args.div),
                   "genome_mbp": args.genome_mbp, "contexts_per_gpu": max(1, args.contexts), "options": "-BW 5 -G 50 -H 650 -M 25 -X 25 -L 15",
                   "reads_per_gpu": n_reads, "read_len": args.read_len, "parallelism": "reads sharded x%d, index replicated, no collective; %d contexts (batches in flight) per GPU" % (world, max(1, args.contexts))},
        "roofline": {"bound": "hbm", "kernel": kname, "unshared": ({"kernel_ms_per_launch": unshared_rows_ms, "achieved": kbytes / (unshared_rows_ms * 1e-3) / 1e9, "frac": kbytes / (unshared_rows_ms * 1e-3) / 8.0e12,
                                                                   "note": "the same launch with one context on the GPU (two steps after the timed region): what rocprofv3 reports for the kernel in profiles/*_one_context.csv"} if (rows_ms > 0 and unshared_rows_ms) else None), "kernel_variant": ("k_ext_rows_pk (packed 16-bit, two cells per instruction)" if (rows_ms > 0 and packed) else ("k_ext_rows (32-bit)" if rows_ms > 0 else kname)), "achieved": (ach_un if ach_un else achieved), "peak": 8000.0, "unit": "GB/s", "frac": (ach_un if ach_un else achieved) / 8000.0, "traffic": traffic,
                     "frac_basis": ("one context on the GPU (full-chip launch), measured in this run after the timed region: what profiles/*_one_context.csv reproduces" if ach_un else "in the timed region"),
                     "frac_unshared": (ach_un / 8000.0 if ach_un else None), "kernel_ms_unshared": unshared_rows_ms, "achieved_in_run": achieved, "frac_in_run": achieved / 8000.0, "kernel_ms_in_run": kernel_ms,
                     "trace_bytes_useful_frac": (tv["records_visitable"] / tv["records_written"] if (tv and tv["records_written"]) else None),
                     "trace_bytes_written": (tv["records_written"] * tv["record_bytes"] if tv else None), "trace_bytes_visitable": (tv["records_visitable"] * tv["record_bytes"] if tv else None),
                     "trace_calls": (tv["calls"] if tv else None), "trace_calls_walking": (tv["walkers"] if tv else None), "whole_path_traffic": ((pmc or {}).get("whole_path") or {}).get("hbm_bytes_per_step"),
                     "algorithmic_bytes_per_launch": kbytes, "algorithmic_bytes_per_read": B, "reads_per_launch": n_reads, "kernel_ms_per_launch": kernel_ms,
                     "kernel_ms_hip_events": rows_ms, "kernel_ms_device_clock": rows_dev_ms,
                     "kernel_stream_bytes_per_launch": stream_bytes, "kernel_stream_frac": (stream_bytes / (kernel_ms * 1e-3) / 8.0e12) if (stream_bytes and kernel_ms > 0) else None,
                     "cell_updates_per_s": counters["dp_ext_cells"] / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0,
                     "valu_frac_nominal": (valu * 2.0 / simd_cycles) if (valu and kernel_ms > 0) else None,
                     "valu_frac_measured_mix": (valu * 4.0 / simd_cycles) if (valu and kernel_ms > 0) else None,
                     "note": "a rows launch that shares the device with other batches takes 3/5 of the workgroups that fit and only one runs at a time (the others' latency-bound kernels run beside it): in the timed region it lasts longer BY DESIGN (`frac_in_run`, `kernel_ms_in_run`); `frac` / `achieved` are the same launch with the device to itself (`frac_unshared`, = the profiles' one-context figure) whenever that was measured; integer DP: `frac` follows SURVEY 8(d) (algorithmic bytes of the reads of one launch / kernel time / 8 TB/s) and is ~1% by construction; the kernel is bound by vector-ALU issue: valu_frac_nominal prices each wave64 VALU instruction at the nominal 2 SIMD cycles, valu_frac_measured_mix at what its mix measures: 4.4 cycles for the packed 16-bit kernel, 4.0 for the 32-bit one (DESIGN.md section 7); kernel_stream_frac counts the trace cells the kernel writes (an implementation choice, not algorithmic traffic)",
                     "pmc": ({k2: pmc[k2] for k2 in ("valu_insts_per_launch", "fetch_bytes_per_launch", "write_bytes_per_launch", "lds_bank_conflict_frac", "gpu_busy_cycles", "git_head", "kernel_source_sha16", "source") if k2 in pmc} if pmc else None)},
        "path": {"algorithmic_bytes_per_read": B, "hbm_frac_whole_path": value * B / (8.0e12 * world),
                 "dp_cell_updates_per_s": (counters["dp_ext_cells"] + counters["dp_gap_cells"]) * steps * world / dt},
        "stage_ms_per_step": {k2: v / steps for k2, v in stage_ms.items()},
        "per_read": {k2: counters[k2] / n_reads for k2 in ("hits", "fragments", "clumps_formed", "clumps_scored", "dp_ext_calls", "dp_ext_rows", "dp_ext_cells", "dp_gap_calls", "dp_gap_rows", "dp_gap_cells", "splits", "ops_out", "ref_bases_touched")},
        "pcie": {"upload_s": t_up, "collect_s": t_down, "clumps": n_clumps},
    }
    if d2h:
        out["value_with_d2h"] = d2h["value_with_d2h"]; out["d2h"] = d2h
        if "value_with_postfilter" in d2h:
            out["value_with_postfilter"] = d2h["value_with_postfilter"]
    if world == 1 and not args.no_extras and not args.quick_extras:
        # the same step on BASELINE's other read lengths (configs 1 and 3), and the whole command line (config 2 end to end)
        wl = []
        for label, n, length, div, nsteps in (("c1: 100 bp reads, r=0.02 (realised 0.7%)", 65536, 100, 0.007, 4), ("c3: 10 kbp reads, r=0.10 (realised 3.4%)", 1024, 10000, 0.034, 3)):
            try:
                wl.append(side_workload(ya, idx, make_reads(cache, fa, "side", n, length, div, 2000), n, local, args.contexts, nsteps, label))
            except Exception as e:
                wl.append({"workload": label, "error": str(e)[:200]})
        # the headline's int32 twin: the same batch with the X-drop extension rows in the 32-bit kernels (the reference computes in int; the packed 16-bit kernel is a
        # lossless narrowing of the same recurrence, bit-identical results -- tests/test_gpu_parity.py::test_both_extension_kernel_families)
        os.environ["YGPU_EXT32"] = "1"
        try:
            w32 = side_workload(ya, idx, reads_path, n_reads, local, args.contexts, max(3, steps // 2), "c2 with the 32-bit extension kernels (YGPU_EXT32=1), dtype int32", blocks=max(1, args.blocks))
            w32["dtype"] = "int32"; wl.append(w32); out["value_int32"] = w32["reads_per_s"]; out["ms_per_step_int32"] = w32["ms_per_step"]; out["ms_per_step_int32_min_max"] = [w32["ms_per_step_min"], w32["ms_per_step_max"]]
        except Exception as e:
            wl.append({"workload": "c2 int32", "error": str(e)[:200]})
        finally:
            del os.environ["YGPU_EXT32"]
        try:
            wl.append(config5_leg(ya, idx, fa, cache, local, args.contexts, "g%dm" % args.genome_mbp))
        except Exception as e:
            wl.append({"workload": "c5", "error": str(e)[:300]})
        wl.insert(1, {"workload": "c2: 1 kbp reads, r=0.05 (realised 1.7%) = the headline", "reads_per_step": n_reads, "steps": steps, "reads_per_s": value, "bases_per_s": value * Lq, "ms_per_step": 1e3 * dt / steps,
                      "k_ext_rows_ms_per_step": rows_dev_ms})
        out["workloads"] = wl
        for w in wl:      # (flat copies: the driver's record drops nested lists)
            tag = (w.get("workload") or "")[:2]
            if tag in ("c1", "c3", "c5") and w.get("reads_per_s"):
                out["value_" + tag] = w["reads_per_s"]
                if w.get("verified") is not None: out["verified_" + tag] = bool(w["verified"].get("identical")) if isinstance(w["verified"], dict) else bool(w["verified"])
        try:
            out["end_to_end"] = end_to_end(ya, idx, fa, cache, args.e2e_reads, 3000)
            out["e2e_reads_per_s"] = out["end_to_end"]["e2e_reads_per_s"]; out["steady_reads_per_s"] = out["end_to_end"]["steady_reads_per_s"]
            out["contexts_up_ms"] = out["end_to_end"].get("contexts_up_ms")
            if args.e2e_reads >= 262144:     # BASELINE config 3's shape through the command line as well: 32 768 reads of 10 kbp (20 batches of ~16 M bases), r = 0.10 (realised 3.4 %)
                out["end_to_end_c3"] = end_to_end(ya, idx, fa, cache, 32768, 3100, read_len=10000, div=0.034, settle_s=40.0)
                out["e2e_c3_reads_per_s"] = out["end_to_end_c3"].get("e2e_reads_per_s"); out["steady_c3_reads_per_s"] = out["end_to_end_c3"].get("steady_reads_per_s")
        except Exception as e:
            out["end_to_end"] = {"error": str(e)[:200]}
        try:
            out["host_ceiling"] = host_ceiling(ya, idx, os.path.join(cache, "e2e_n%d_l%d_s%d.fa" % (args.e2e_reads, 1000, 3000)))
            out["host_ceiling_reads_per_s"] = out["host_ceiling"]["host_ceiling_reads_per_s"]
        except Exception as e:
            out["host_ceiling"] = {"error": str(e)[:300]}
        if args.e2e_reads >= 262144:
            try:
                out["eight_logical_devices"] = eight_logical_devices(ya, cache, args.seed, args.e2e_reads)
            except Exception as e:
                out["eight_logical_devices"] = {"error": str(e)[:300]}
    if world > 1 and not args.no_extras:
        # N > 1: the whole command line over the N devices (`yaha -gpus N`, one process, N x 3 contexts, index image uploaded to every device), the other ranks idle
        try:
            out["end_to_end"] = end_to_end(ya, idx, fa, cache, args.e2e_reads, 3000, gpus=min(world, max(1, torch.cuda.device_count())))
            out["e2e_reads_per_s"] = out["end_to_end"]["e2e_reads_per_s"]; out["steady_reads_per_s"] = out["end_to_end"]["steady_reads_per_s"]
        except Exception as e:
            out["end_to_end"] = {"error": str(e)[:300]}
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(idx, fa, reads_path, n_reads, cache, args.cpu_seconds, args.read_len, args.div)
        except Exception as e:  # the baseline is a reported number, never a reason to lose the measurement
            out["cpu_baseline"] = {"value": None, "unit": "reads/s", "cores": os.cpu_count(), "kind": "error", "sample": str(e)[:200]}
        cb = out["cpu_baseline"]
        if cb.get("kind") == "reference" and cb.get("reference_sam"):
            try:
                e2e_path = os.path.join(cache, "e2e_n%d_l%d_s%d.fa" % (args.e2e_reads, args.read_len, 3000))
                out["verified"] = verify_against_reference(ya, idx, cb, e2e_path if not args.no_extras else None, cache)
            except Exception as e:
                out["verified"] = {"identical": False, "error": str(e)[:300]}
            for k2 in ("sample_path", "reference_sam"):
                cb.pop(k2, None)
            v = out["verified"]
            out["verified_identical"] = bool(v.get("identical")); out["verified_reads"] = int(v.get("reads") or 0) + int((v.get("e2e_sample") or {}).get("reads") or 0)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier(group=cpu_group)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
