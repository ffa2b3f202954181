"""The N>1 path on CPU: unit partitioning and the timing gather over torch.distributed (gloo, world_size 2)."""
import os
import socket
import subprocess
import sys

import numpy as np

from slowflow_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_covers_every_unit_once():
    for n in (0, 1, 7, 16, 128, 129):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                lo, hi = shard.partition(n, world, r)
                assert 0 <= lo <= hi <= n
                seen += list(range(lo, hi))
            assert seen == list(range(n))
            sizes = [shard.partition(n, world, r)[1] - shard.partition(n, world, r)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert shard.windows_of_sequence(2) == [(0, False), (0, True), (1, False), (1, True)]


WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from slowflow_amd import shard
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n = 11
lo, hi = shard.partition(n, world, rank)
local = {i: 0.5 + i + 100 * rank for i in range(lo, hi)}
full = shard.gather_timings(dist, local, n)
mx = shard.max_over_ranks(dist, 1.0 + rank)
dist.barrier()
if rank == 0:
    print("RESULT", ",".join("%%.1f" %% v for v in full), mx)
dist.destroy_process_group()
'''


def test_gather_timings_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()
    vals = [float(v) for v in line[1].split(",")]
    lo1, _ = shard.partition(11, 2, 1)
    assert vals == [0.5 + i + (100 if i >= lo1 else 0) for i in range(11)]
    assert float(line[2]) == 2.0
