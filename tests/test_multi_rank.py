"""CPU tier, world_size 2 over gloo: the N>1 path.  Reads are independent units: ranks take
disjoint shards, run the hot path on their shard with the index replicated, and exchange nothing on the data path;
only the barrier and the max-over-ranks timing go through torch.distributed.  Here the oracle stands in for the
device (no GPU in this tier); what is checked is the sharding / ordering / reduction logic that bench.py and the
CLI's `-gpus N` rely on: the union of the per-rank outputs, re-ordered by batch ticket, equals the single-rank output."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import golden_lines, strip_pg


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank_main(rank, world, port, index, reads, batch, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import oracle
    import yaha_amd as ya
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    parts = {}
    with ya.Session(["-x", index, "-q", reads, "-osh", "stdout"]) as s:
        header = s.header()
        ticket = 0
        while True:
            b = s.next_batch(batch)                  # every rank reads the stream; batches are dealt round-robin by ticket
            if b.n_reads == 0:
                break
            if bench.owner_of(ticket, world) == rank:
                r, _own = oracle.run(s.index, s.params, b, threads=2)
                parts[ticket] = s.emit(r)
            ticket += 1
    dist.barrier()
    t = bench.max_over_ranks(float(rank + 1), dist, device="cpu")
    gathered = [None] * world
    dist.all_gather_object(gathered, parts)
    if rank == 0:
        text = header + bench.merge_by_ticket(gathered)
        open(os.path.join(out_dir, "merged.sam"), "w").write(text)
        open(os.path.join(out_dir, "tmax.txt"), "w").write(str(t))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_and_merge(work, index11, tmp_path):
    port = _free_port()
    mp.spawn(_rank_main, args=(2, port, index11, os.path.join(work, "rchim.fa"), 37, str(tmp_path)), nprocs=2, join=True)
    merged = strip_pg(open(tmp_path / "merged.sam").read())
    assert merged == golden_lines("rchim_default")
    assert float(open(tmp_path / "tmax.txt").read()) == 2.0


def test_bench_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the driver's form) must start two ranks itself -- as child processes, the parent never touches a
    GPU -- and relay rank 0's single JSON line with n_gpus == 2.  YAHA_BENCH_STUB=1 swaps the device step for a sleep and RCCL for gloo: launch path only."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["YAHA_BENCH_STUB"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().split("\n") if l.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["steps"] == 3 and j["warmup"] == 1
    assert j["ms_per_step"] >= 4.0                      # the max over ranks (rank 1 sleeps 4 ms a step), not rank 0's 2 ms
    # under an external launcher (WORLD_SIZE set) the same file is a rank, not a launcher: one rank, no spawn
    env1 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2"], env=env1, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p1.returncode == 0 and json.loads(p1.stdout.decode().strip())["n_gpus"] == 1
