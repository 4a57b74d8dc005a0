"""The N > 1 path on CPU: two gloo ranks each own a shard; one all-reduce of the per-step sums per flush."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from tests import cases

torch.set_default_dtype(torch.float64)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, nt, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from oracle import sc_oracle as orc
    from semiclassical_amd import distributed as D
    torch.set_default_dtype(torch.float64)
    torch.set_num_threads(1)
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = cases.load("hk_as5_chi002")
    n_total = g["zi"].shape[1]
    counts = [D.shard_count(n_total, i, world) for i in range(world)]
    lo = sum(counts[:rank])
    sl = slice(lo, lo + counts[rank])
    # this rank's shard of the SAME initial conditions, normalised with the global N
    pot = cases.oracle_potential(g)
    prop = orc.HKOracle(cases.T(g["Gamma_i"]), cases.T(g["Gamma_t"]))
    prop.set_initial_conditions(cases.T(g["q0"]), cases.T(g["p0"]), cases.T(g["Gamma_0"]),
                                cases.T(g["zi"][:, sl]), cases.T(g["probi"][sl]))
    prop.ntraj = n_total                      # Monte-Carlo weight 1/(N_total P)
    slots = torch.zeros((nt, 5))
    for t in range(nt):
        c = torch.sum(prop.autocorrelation_qp() / prop._mc_weight())
        slots[t, 0], slots[t, 1] = c.real, c.imag
        slots[t, 4] = rank + 1.0              # rank-local column must survive the flush
        prop.step(pot, float(g["dt"]))
    D.flush_correlations(slots)
    assert slots[0, 4] == rank + 1.0
    if rank == 0:
        q.put(slots[:, :2].numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_flush_equals_single_process():
    nt, world = 12, 2
    g = cases.load("hk_as5_chi002")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, nt, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from semiclassical_amd import hostmath
    phase = np.exp(1j * hostmath.time_grid(nt, float(g["dt"])) * float(g["E0"]))
    cauto = (out[:, 0] + 1j * out[:, 1]) * phase
    assert cases.rel_err(cauto, g["cauto"][:nt]) < 1e-12


def test_shard_counts():
    from semiclassical_amd import distributed as D
    assert [D.shard_count(10, r, 4) for r in range(4)] == [3, 3, 2, 2]
    assert sum(D.shard_count(10 ** 6, r, 8) for r in range(8)) == 10 ** 6
    assert D.shard_count(5, 0, 1) == 5


def test_flush_is_identity_without_process_group():
    from semiclassical_amd import distributed as D
    s = torch.arange(10.0).reshape(2, 5)
    assert torch.equal(D.flush_correlations(s.clone()), s)
