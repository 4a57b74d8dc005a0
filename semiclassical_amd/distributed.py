"""Sharding of trajectory batches over the GPUs of one node and the per-flush all-reduce.

Trajectories are independent; the only coupling is the sum over trajectories in C_auto(t) and k_ic(t)
(reference propagators.py:837, 909).  Every rank integrates its own shard for the whole repetition and
accumulates the raw sums of each time step in a device buffer ``slots`` (nt, 5).  Because each term
already carries the weight 1/(N_total P(qi,pi)), the global correlation functions are the plain SUM of
the per-rank buffers: one all-reduce of 4*nt doubles per flush (RCCL over xGMI when the process group
uses the ``nccl`` backend; ``gloo`` for tests, CUDA tensors are then staged through the host).  No other
collective exists on the path.

Process model: one process per GPU.  ``launch_local_ranks`` starts the rank processes of one node from a
parent that never touches the GPU (it only counts to N and waits); ``init_from_env`` is what every rank
calls first.  This module does not import the HIP engine at import time (``RcclCommunicator`` binds it when
it is constructed, inside a rank process).

Two transports for the flush:
  * a ``torch.distributed`` process group (``nccl`` = RCCL on ROCm, or ``gloo``): ``flush_correlations(slots)``;
  * the C-ABI's own binding of librccl, ``sc_flush_allreduce`` (include/semiclassical_hip.h), for consumers
    without torch.distributed: ``flush_correlations(slots, comm=RcclCommunicator(...))`` -- the only thing
    that crosses processes on the host side is RCCL's 128-byte unique id.
"""
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun or launch_local_ranks);
    returns (rank, world, local_rank)"""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = os.environ.get("SC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_environment(rank, world, port, base=None):
    """environment of rank `rank` of a single-node job: what torchrun would export"""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # dmabuf IPC: the build image's environment notes say this host driver supports only dmabuf IPC and that RCCL /
    # device-memory sharing across processes fails with "hipIpcGetMemHandle: invalid argument" without it (the image exports
    # it already; kept for ranks started from a scrubbed environment).  Not verified here: the boxes have one GPU.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def launch_local_ranks(argv, nproc, timeout=None, extra_env=None):
    """Start ``nproc`` copies of ``python argv...`` as ranks 0..nproc-1 of one node and wait for them.

    The caller must not have initialised the GPU: the children are fresh interpreters (plain fork+exec of
    ``sys.executable``), rank r binds to cuda:r itself.  stdout / stderr are inherited, so rank 0's single
    JSON line is the job's output.  Returns the largest exit code; on a failing or hung rank the others are
    terminated (by PID).
    """
    port = free_port()
    procs = []
    for r in range(nproc):
        env = rank_environment(r, nproc, port)
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env))
    rc = 0
    deadline = None if timeout is None else time.monotonic() + timeout
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = max(rc, code if code > 0 else 1)
                    for o in pending:            # one rank died: the collective of the others would hang
                        o.terminate()
            if pending and deadline is not None and time.monotonic() > deadline:
                rc = max(rc, 124)
                for o in pending:
                    o.terminate()
                deadline = None
            if pending:
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def shard_count(n_total, rank, world):
    """trajectories owned by `rank`: equal shares, the first (n_total % world) ranks take one more"""
    base, rem = divmod(int(n_total), int(world))
    return base + (1 if rank < rem else 0)


def shard_slice(n_total, rank, world):
    """the contiguous index range of `rank`'s shard"""
    lo = sum(shard_count(n_total, r, world) for r in range(rank))
    return slice(lo, lo + shard_count(n_total, rank, world))


class RcclCommunicator(object):
    """An RCCL communicator owned through the C-ABI (``sc_comm_init`` / ``sc_flush_allreduce``): the flush without
    torch.distributed.  Collective constructor -- every rank of the job builds one on ITS device.

    The unique id travels over a host channel: ``store`` is anything with ``set(key, bytes)`` / ``get(key) -> bytes``
    (a ``torch.distributed.TCPStore``, which needs no process group); without one, an initialised process group
    broadcasts it, or a TCPStore is opened on MASTER_ADDR : MASTER_PORT + 1.
    """

    def __init__(self, rank, world, device, store=None, key="sc_rccl_unique_id"):
        import ctypes
        from ._lib import lib, check
        self._lib, self._check = lib, check
        self.rank, self.world, self.device = int(rank), int(world), torch.device(device)
        uid = ctypes.create_string_buffer(128)
        if self.rank == 0:
            check(lib.sc_comm_unique_id(uid))
        if self.world > 1:
            uid = ctypes.create_string_buffer(self._exchange(bytes(uid.raw), store, key), 128)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.sc_comm_init(uid, self.world, self.rank, ctypes.byref(handle)))
        self.handle = handle

    def _exchange(self, mine, store, key):
        if store is None and dist.is_available() and dist.is_initialized():
            box = [mine if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        if store is None:
            store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29511")) + 1,
                                  self.world, is_master=(self.rank == 0))
            self._store = store                                     # the server side has to outlive the clients' get()
        if self.rank == 0:
            store.set(key, mine)
            return mine
        return bytes(store.get(key))

    def all_reduce_sum(self, t):
        """sum of the contiguous float64 device tensor ``t`` over the ranks, in place, on torch's current stream"""
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        stream = torch.cuda.current_stream(t.device).cuda_stream
        import ctypes
        self._check(self._lib.sc_flush_allreduce(ctypes.c_void_p(t.data_ptr()), t.numel(), self.handle, stream))
        return t

    def rank_count(self):
        import ctypes
        r, n = ctypes.c_int32(), ctypes.c_int32()
        self._check(self._lib.sc_comm_rank_count(self.handle, ctypes.byref(r), ctypes.byref(n)))
        return r.value, n.value

    def destroy(self):
        if getattr(self, "handle", None):
            self._check(self._lib.sc_comm_destroy(self.handle))
            self.handle = None


def flush_correlations(slots, group=None, comm=None):
    """sum the raw per-step correlation sums over all ranks, in place (columns 0..3 of ``slots``).

    Column 4 (reserved) is left rank-local.  A single collective per call.  With the ``gloo`` backend a
    device tensor is staged through the host (gloo reduces host memory); with ``nccl`` (RCCL) the
    reduction runs on the device buffers directly.  ``comm``: an ``RcclCommunicator`` -- the same single
    all-reduce through the C-ABI's ``sc_flush_allreduce`` instead of torch.distributed.
    """
    if comm is not None:
        buf = slots[:, :4].contiguous()
        comm.all_reduce_sum(buf)
        slots[:, :4] = buf
        return slots
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return slots
    buf = slots[:, :4].contiguous()
    if buf.is_cuda and dist.get_backend(group) == "gloo":
        host = buf.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        buf.copy_(host)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    slots[:, :4] = buf
    return slots


class _Local(object):
    """sentinel group: "this rank only" (the default of the O(n^2) diagnostics, which must not turn into collectives
    just because a process group exists)"""

    def __repr__(self):
        return "LOCAL"


LOCAL = _Local()


def _active(group=None):
    if group is LOCAL:
        return False
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def world_size(group=None):
    return dist.get_world_size(group) if _active(group) else 1


def get_rank(group=None):
    return dist.get_rank(group) if _active(group) else 0


def broadcast_object(obj, src=0, group=None):
    """``obj`` of rank ``src`` on every rank (host objects: seeds, small configuration)"""
    if not _active(group):
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]


def all_gather_rows(t, group=None):
    """the rows (first axis) of ``t`` of all ranks, concatenated in rank order, on ``t``'s device.  Shards may differ in
    size.  Under ``gloo`` device tensors are staged through the host (as in flush_correlations)."""
    if not _active(group):
        return t
    world = dist.get_world_size(group)
    stage = t.is_cuda and dist.get_backend(group) == "gloo"
    src = t.contiguous().cpu() if stage else t.contiguous()
    counts = [torch.zeros(1, dtype=torch.int64, device=src.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([src.shape[0]], dtype=torch.int64, device=src.device), group=group)
    counts = [int(c.item()) for c in counts]
    width = max(counts)
    pad = torch.zeros((width,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    pad[:src.shape[0]] = src
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
    return out.to(t.device) if stage else out


def all_reduce_sum(t, group=None):
    """sum of ``t`` over the ranks, in place (staged through the host under gloo)"""
    if not _active(group):
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
