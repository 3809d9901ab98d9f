"""GPU tier: the N > 1 path of bench.py on the box's one GPU -- `python bench.py --gpus 2` starts two ranks (child processes of a parent that never touches a GPU), both
take GPU 0 and torch.distributed runs over gloo (YAHA_BENCH_BACKEND=gloo: RCCL refuses two ranks on one device), each rank times its own contexts on its own reads, the
maximum over the ranks goes into the one JSON line.  Everything of the multi-GPU run but RCCL itself: launch, rendezvous, per-rank inputs, barriers, reduction, the
CPU-side waits behind the timed region, rank 0's command-line leg.  Small genome (100 Mbp, -L 15), one context a rank, full batches of 16 384 reads: two processes running
the same kernels in lock step on one device is also what showed that the single-pass scans' tiles must be tickets (seed.h: tileTicket) -- with tile = blockIdx the two
fragment scans stalled each other here."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_on_one_gpu(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(YAHA_BENCH_BACKEND="gloo", YAHA_BENCH_CACHE=os.environ.get("YAHA_BENCH_CACHE", str(tmp_path / "cache")))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--genome-mbp", "100", "--contexts", "1", "--reads-per-gpu", "16384",
                        "--e2e-reads", "16384"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().split("\n") if l.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "weak" and j["value"] > 0
    assert "x2" in j["config"]["parallelism"]
    e = j["end_to_end"]
    assert "error" not in e and e["sam_records"] > 16000 and e["gpus"] == 1          # the command-line leg ran over the devices that exist
