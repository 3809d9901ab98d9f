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
reads that is already resident in HBM; results stay in HBM.  Every rank drives --contexts (default 2) device contexts on its
GPU, one host thread each, which take the K timed steps from a common counter: two batches are in flight per GPU, so that one
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
        subprocess.check_call([sim, "genome", "--seed", str(seed), "--out", fa, "--seqs", "24", "--len", str(genome_mbp * 1000000), "--repeat-frac", "0.45", "--nrun", "6", "--lowcomplex", "8"])
        ya.build_index(["-g", fa, "-L", "15"])
        open(done, "w").write("ok")
        log("built genome + index in %.1fs" % (time.time() - t))
    return fa, idx


def make_reads(cache, fa, tag, n, length, div, seed):
    path = os.path.join(cache, "%s_n%d_l%d_s%d.fa" % (tag, n, length, seed))
    if not os.path.exists(path):
        tmp = path + ".tmp%d" % os.getpid()
        subprocess.check_call([os.path.join(ROOT, "tools", "yaha_sim"), "reads", "--genome", fa, "--out", tmp, "--seed", str(seed), "--n", str(n), "--len", str(length), "--div", str(div)])
        os.replace(tmp, path)
    return path


def head_reads(src, dst, n):
    with open(src) as f, open(dst, "w") as o:
        k = 0
        for line in f:
            if line.startswith(">"):
                k += 1
                if k > n:
                    break
            o.write(line)


def cpu_baseline(idx, reads_path, n_reads, cache, target_s):
    """reference yaha -t <all cores> on a bounded sample; wall time minus a zero-work run (index mmap pre-touch)."""
    import oracle
    cores = os.cpu_count() or 1
    if oracle.have_reference():
        one = os.path.join(cache, "one_read.fa")
        head_reads(reads_path, one, 1)
        t = time.time(); oracle.run_reference(["-x", idx, "-q", one, "-osh", "/dev/null", "-t", str(cores)]); t_zero = time.time() - t
        probe = os.path.join(cache, "probe.fa")
        n_probe = min(n_reads, 4 * cores)
        head_reads(reads_path, probe, n_probe)
        t = time.time(); oracle.run_reference(["-x", idx, "-q", probe, "-osh", "/dev/null", "-t", str(cores)]); t_probe = max(time.time() - t - t_zero, 1e-3)
        n = int(min(n_reads, max(n_probe, target_s * n_probe / t_probe)))
        sample = os.path.join(cache, "sample.fa")
        head_reads(reads_path, sample, n)
        t = time.time(); oracle.run_reference(["-x", idx, "-q", sample, "-osh", "/dev/null", "-t", str(cores)]); dt = max(time.time() - t - t_zero, 1e-3)
        return {"value": n / dt, "unit": "reads/s", "cores": cores, "kind": "reference",
                "sample": "%d of the same 1 kbp reads, oracle/_ref/yaha -t %d, %.1fs wall minus %.1fs zero-read run" % (n, cores, dt + t_zero, t_zero)}
    import yaha_amd as ya
    with ya.Session(["-x", idx, "-q", reads_path]) as s:
        b = s.next_batch(min(n_reads, 64 * cores))
        t = time.time(); oracle.run(s.index, s.params, b, threads=cores); dt = time.time() - t
        return {"value": b.n_reads / dt, "unit": "reads/s", "cores": cores, "kind": "port", "sample": "%d reads, oracle/hotpath.cpp on %d threads (hot path only)" % (b.n_reads, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads-per-gpu", type=int, default=16384)
    ap.add_argument("--read-len", type=int, default=1000)
    ap.add_argument("--genome-mbp", type=int, default=100)
    ap.add_argument("--div", type=float, default=0.017)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--contexts", type=int, default=2, help="device contexts (batches in flight) per GPU")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)
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
        fa, idx = ensure_inputs(cache, args.genome_mbp, args.seed)
    barrier()
    fa, idx = ensure_inputs(cache, args.genome_mbp, args.seed)
    reads_path = make_reads(cache, fa, "g%dm" % args.genome_mbp, args.reads_per_gpu, args.read_len, args.div, 1000 + rank)

    import threading
    with ya.Session(["-x", idx, "-q", reads_path]) as s:
        b = s.next_batch(args.reads_per_gpu)
        n_reads = b.n_reads
        offs = C.cast(b.offsets, C.POINTER(C.c_uint64))
        n_bases = int(offs[n_reads] - offs[0])
        # args.contexts device contexts on this GPU (they share the index image), one host thread each; every step is one pass of
        # the whole hot path over the batch by ONE context, and the contexts take the K steps from a common counter: while one
        # is in a latency-bound stage the other one's kernels fill the device.
        ctxs = [ya.Context(s.index, s.params, device=local)]
        for _ in range(1, max(1, args.contexts)):
            ctxs.append(ya.Context(s.index, s.params, device=local, parent=ctxs[0]))
        t = time.time()
        for c in ctxs:
            c.upload(b)
        t_up = (time.time() - t) / len(ctxs)
        for c in ctxs:
            for _ in range(args.warmup):
                c.run()
        stage_ms = {}
        lock = threading.Lock()
        todo = [args.steps]

        def stepper(c):
            while True:
                with lock:
                    if todo[0] <= 0:
                        return
                    todo[0] -= 1
                c.run()                                     # synchronous: returns when the results are complete in HBM
                tm = c.timing()[1]
                with lock:
                    for k, v in tm.items():
                        stage_ms[k] = stage_ms.get(k, 0.0) + v
        barrier()
        t0 = time.time()
        th = [threading.Thread(target=stepper, args=(c,)) for c in ctxs]
        for x in th:
            x.start()
        for x in th:
            x.join()
        barrier()
        dt = time.time() - t0
        t = time.time(); r = ctxs[0].collect(); t_down = time.time() - t
        counters = r.counters.as_dict()
        n_clumps = int(r.n_clumps)
        for c in reversed(ctxs):
            c.close()
    dt = max_over_ranks(dt, dist)
    if rank != 0:
        if dist is not None:
            dist.barrier()
        return

    steps = args.steps
    total_reads = world * n_reads * steps
    value = total_reads / dt
    # algorithmic bytes per read, SURVEY.md 8(d): read codes + 2 strands x 2 u32 table words per k-mer + hit words +
    # touched reference nibbles + result records
    k = s.params.wordLen
    Lq = n_bases / n_reads
    B = Lq + 16 * (Lq - k + 1) + 4 * counters["hits"] / n_reads + counters["ref_bases_touched"] / n_reads / 2 + (24 * counters["clumps_scored"] + 3 * counters["ops_out"]) / n_reads
    align_ms = stage_ms.get("align_dp", 0.0) / steps
    # Dominant kernel (profiles/): k_ext_rows, the X-drop extension rows, one problem per lane.  Its ALGORITHMIC bytes per launch
    # (DESIGN.md section 5): per computed row 1 query code + half a byte of packed reference + 12 B of trace cells (21 x 4 bit),
    # per problem a 16-byte descriptor and a 32-byte result.  Duration = HIP events around its launch(es) on its stream.
    rows_ms = stage_ms.get("ext_rows", 0.0) / steps
    rows_dev_ms = stage_ms.get("ext_rows_device_clock", 0.0) / steps
    if rows_ms > 0:
        # The launch is bracketed by HIP events on its stream, and the kernel also stamps wall_clock64() at its first wave's start and
        # last wave's end.  With two contexts per GPU the event bracket additionally contains the time the launch waits behind the
        # other context's kernels, so the device-clock duration (the one rocprofv3 reports for the kernel) is the one used.
        kname, kernel_ms = "k_ext_rows", (rows_dev_ms if rows_dev_ms > 0 else rows_ms)
        kbytes = 13.5 * counters["dp_ext_rows"] + 48.0 * counters["dp_ext_calls"]
    else:                                                    # other band widths run the wave-per-root kernel
        kname, kernel_ms = "k_align", align_ms
        kbytes = B * n_reads
    achieved = (kbytes / (kernel_ms * 1e-3)) / 1e9 if kernel_ms > 0 else 0.0
    traffic, pmc = None, None
    pj = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pj):
        try:
            pmc = json.load(open(pj))
            if pmc.get("reads_per_gpu") == n_reads and pmc.get("kernel") == kname:
                traffic = pmc.get("hbm_bytes_per_launch")
        except Exception:
            traffic, pmc = None, None
    out = {
        "metric": "aligned reads/s (whole node), 1 000 bp reads, OQC mode hot path", "value": value, "unit": "reads/s",
        "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "bases_per_s": value * Lq,
        "config": {"workload": "synthetic hg18-like genome %d Mbp (24 seqs, 45%% repeats), index -L 15 -S 1 -H 65525, %d x %d bp reads per GPU at %.1f%% divergence, defaults -BW 5 -G 50 -H 650 -M 25 -X 25, hot path A1..A10 with inputs resident in HBM"
                   % (args.genome_mbp, n_reads, args.read_len, 100 * args.div),
                   "reads_per_gpu": n_reads, "read_len": args.read_len, "parallelism": "reads sharded x%d, index replicated, no collective; %d contexts (batches in flight) per GPU" % (world, max(1, args.contexts))},
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                     "algorithmic_bytes_per_launch": kbytes, "kernel_ms_per_launch": kernel_ms,
                     "kernel_ms_hip_events": rows_ms, "kernel_ms_device_clock": rows_dev_ms,
                     "cell_updates_per_s": counters["dp_ext_cells"] / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0,
                     "note": "integer DP: the kernel is bound by vector-ALU issue, not HBM (see DESIGN.md section 6 and profiles/)",
                     "pmc": ({k2: pmc[k2] for k2 in ("valu_insts_per_launch", "valu_cycles_per_inst", "valu_issue_frac", "fetch_bytes_per_launch", "write_bytes_per_launch", "source") if k2 in pmc} if pmc and traffic is not None else None)},
        "path": {"algorithmic_bytes_per_read": B, "hbm_frac_whole_path": value * B / (8.0e12 * world),
                 "dp_cell_updates_per_s": (counters["dp_ext_cells"] + counters["dp_gap_cells"]) * steps * world / dt},
        "stage_ms_per_step": {k2: v / steps for k2, v in stage_ms.items()},
        "per_read": {k2: counters[k2] / n_reads for k2 in ("hits", "fragments", "clumps_formed", "clumps_scored", "dp_ext_calls", "dp_ext_rows", "dp_ext_cells", "dp_gap_calls", "dp_gap_rows", "dp_gap_cells", "splits", "ops_out", "ref_bases_touched")},
        "pcie": {"upload_s": t_up, "collect_s": t_down, "clumps": n_clumps},
    }
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(idx, reads_path, n_reads, cache, args.cpu_seconds)
        except Exception as e:  # the baseline is a reported number, never a reason to lose the measurement
            out["cpu_baseline"] = {"value": None, "unit": "reads/s", "cores": os.cpu_count(), "kind": "error", "sample": str(e)[:200]}
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()


if __name__ == "__main__":
    main()
