"""One rank of the engine-level multi-process test (started by semiclassical_amd.distributed.launch_local_ranks).

Every rank runs the HIP engine on ITS shard of a golden case's initial conditions on cuda:0 (two small processes
sharing the one GPU of the test box), with the global N as Monte-Carlo weight, leaves the raw per-step sums on the
device, flushes them through distributed.flush_correlations (gloo here; the same call is the RCCL all-reduce under
nccl) and rank 0 stores the resulting correlation functions.

    python tests/_rank_engine.py CASE NT OUT.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.set_default_dtype(torch.float64)


def main():
    case, nt, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    import torch.distributed as dist
    from semiclassical_amd import distributed as D, propagators as PR
    from tests import cases
    from tests.engine_cases import engine_potential
    rank, world, _ = D.init_from_env()
    dev = torch.device("cuda", int(os.environ.get("SC_TEST_DEVICE", "0")))
    torch.cuda.set_device(dev)
    g = cases.load(case)
    n_total = g["zi"].shape[1]
    sl = D.shard_slice(n_total, rank, world)
    Gi, Gt = cases.T(g["Gamma_i"]), cases.T(g["Gamma_t"])
    if "alpha" in g:
        prop = PR.WaltonManolopoulosPropagator(Gi, Gt, float(g["alpha"]), float(g["beta"]), device=dev)
    else:
        prop = PR.HermanKlukPropagator(Gi, Gt, device=dev)
    prop.set_initial_conditions(cases.T(g["q0"]), cases.T(g["p0"]), cases.T(g["Gamma_0"]),
                                cases.T(g["zi"][:, sl]), cases.T(g["probi"][sl]), ntraj_total=n_total)
    dt, E0 = float(g["dt"]), float(g["E0"])
    slots = torch.zeros((nt, 5), dtype=torch.float64, device=dev)
    prop.run(engine_potential(g), dt, nt, E0, slots=slots)
    slots[:, 4] = rank + 1.0                 # the rank-local column must survive the flush
    D.flush_correlations(slots)
    prop.synchronize()
    assert float(slots[0, 4]) == rank + 1.0
    cauto, kic = prop.finalize_slots(slots, 0.0, dt, E0)
    # the O(n^2) diagnostic across ranks: own bras against the all-gathered kets, one all-reduce (every rank gets the whole norm)
    norm = prop.norm(across_ranks=True) if os.environ.get("SC_TEST_NORM") else float("nan")
    # the default norm() is rank-local: called by ONE rank only it must neither hang nor see the other rank's kets
    local_norm = prop.norm() if (os.environ.get("SC_TEST_NORM") and rank == 0) else float("nan")
    norms = [None] * world
    if world > 1:
        dist.all_gather_object(norms, norm)
    else:
        norms = [norm]
    if rank == 0:
        np.savez(out, cauto=cauto, kic=kic, world=world, backend=dist.get_backend() if world > 1 else "none",
                 norms=np.array(norms, dtype=float), local_norm_rank0=local_norm)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
