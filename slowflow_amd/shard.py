"""Frame-window data parallelism over the GPUs of one node (one process per GPU).

The units of the path are independent: one unit = one (jet, direction) `variational` call (slow_flow.cpp:706-1047 runs
them as independent OpenMP iterations).  They are sharded in contiguous blocks (neighbouring jets share frames); there is
no data-path collective.  The only exchange is the gather of per-unit timings, a few hundred bytes."""
import numpy as np


def partition(n_units, world_size, rank):
    """contiguous block [lo, hi) of `rank`; blocks differ by at most one unit and cover every unit exactly once"""
    lo = n_units * rank // world_size
    hi = n_units * (rank + 1) // world_size
    return lo, hi


def windows_of_sequence(n_jets):
    """the (jet, backward) units of a sequence in the driver's order: forward then backward of every jet"""
    return [(j, b) for j in range(n_jets) for b in (False, True)]


def gather_timings(dist, local_seconds, n_units, device=None):
    """all ranks contribute the timings of their units; every rank receives the full per-unit vector.
    `dist` is torch.distributed (backend nccl = RCCL on the GPUs, gloo on CPU); with dist None (single process) the local
    vector is returned."""
    import torch
    full = torch.zeros(n_units, dtype=torch.float64, device=device)
    for i, s in local_seconds.items():
        full[i] = s
    if dist is not None and dist.is_initialized():       # (also with ONE rank: the collective then runs through RCCL on the one GPU -- the leg a one-GPU box can exercise)
        dist.all_reduce(full, op=dist.ReduceOp.SUM)      # disjoint supports: the sum is the gather
    return full.cpu().numpy()


def max_over_ranks(dist, value, device=None):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value, device=None):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
