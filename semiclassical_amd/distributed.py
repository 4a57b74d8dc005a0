"""Sharding of trajectory batches over the GPUs of one node and the per-flush all-reduce.

Trajectories are independent; the only coupling is the sum over trajectories in C_auto(t) and k_ic(t)
(reference propagators.py:837, 909).  Every rank integrates its own shard for the whole repetition and
accumulates the raw sums of each time step in a device buffer ``slots`` (nt, 5).  Because each term
already carries the weight 1/(N_total P(qi,pi)), the global correlation functions are the plain SUM of
the per-rank buffers: one all-reduce of 4*nt doubles per flush (RCCL over xGMI when the process group
uses the ``nccl`` backend; ``gloo`` on CPU tensors for tests).  No other collective exists on the path.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); returns (rank, world, local_rank)"""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_count(n_total, rank, world):
    """trajectories owned by `rank`: equal shares, the first (n_total % world) ranks take one more"""
    base, rem = divmod(int(n_total), int(world))
    return base + (1 if rank < rem else 0)


def flush_correlations(slots, group=None):
    """sum the raw per-step correlation sums over all ranks, in place (columns 0..3 of ``slots``).

    Column 4 (reserved) is left rank-local.  A single collective per call.
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return slots
    buf = slots[:, :4].contiguous()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    slots[:, :4] = buf
    return slots
