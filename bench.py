#!/usr/bin/env python
"""Headline benchmark: trajectory-steps/s of the HK loop on the synthetic 60-mode anharmonic-AS model.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--ntraj n_per_gpu]

A "step" is one pass of the hot path over one batch: (C_auto, k_ic, RK4 step + prefactor) for all
trajectories (the loop body of reference cli.py:401-436).  Workload = BASELINE.json configs[1]:
anharmonic-AS, D = 60, 10^5 trajectories per GPU, HK, fp64, dt = 0.005 fs.
Inputs are resident in HBM before the timed region; the timed region ends with the single all-reduce
of the accumulated correlation sums (the "flush").  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling


def as60_model(dim=60):
    """synthetic 60-mode AS model, SURVEY.md section 8d config 2 (other `dim` only for the tools/ experiments)"""
    from semiclassical_amd import units
    rng = np.random.default_rng(60)
    omega_cm = np.linspace(160.0, 3300.0, dim)
    S = rng.uniform(0, 0.1, dim) * rng.choice([-1, 1], dim)
    nac = rng.normal(0, 1e-4, dim)
    chi = np.full(dim, 0.02)
    omega = torch.from_numpy(omega_cm / units.hartree_to_wavenumbers)
    S, nac, chi = torch.from_numpy(S), torch.from_numpy(nac), torch.from_numpy(chi)
    q0 = torch.sqrt(2.0 * abs(S) / omega) * torch.sign(S)
    dt = 0.005 / units.autime_to_fs
    return omega, chi, nac, q0, dt


def algorithmic_bytes_per_traj_step(D):
    """read y + write y + per-trajectory side data = 64 D^2 + 64 D + 72"""
    return 64 * D * D + 64 * D + 72


def profiled_traffic(n, dim):
    """HBM bytes per step-kernel launch from the committed rocprofv3 PMC passes (profiles/), if they match this workload"""
    path = os.path.join(ROOT, "profiles", "r1_hbm_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t["workload"]["ntraj"] == n and t["workload"]["dim"] == dim:
            return t["traffic_bytes_per_launch"], "profiles/r1_hbm_traffic.json (FETCH_SIZE calibrated + WRITE_SIZE, separate passes)"
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def cpu_baseline(omega, chi, nac, q0, dt, n=2000, nt=4):
    """the CPU oracle (torch eager restatement of the reference's op sequence) on a bounded sample"""
    from oracle import sc_oracle as orc
    # the GPU box gives one GPU's share of the host: 16 cores (oversubscribing the visible CPUs stalls torch)
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0)), os.cpu_count() or 1))
    G = torch.diag(omega)
    E0 = float(0.5 * omega.sum())
    pot = orc.MorseOracle(omega, chi.clone(), nac)
    prop = orc.HKOracle(G, G)
    torch.manual_seed(0)
    prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n)
    orc.run_loop(prop, pot, dt, 1, E0)                      # warm-up step
    t0 = time.perf_counter()
    orc.run_loop(prop, pot, dt, nt, E0)
    wall = time.perf_counter() - t0
    return {"value": n * nt / wall, "unit": "trajectory-steps/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"D=60 anharmonic-AS, n={n}, {nt} steps after 1 warm-up step, {wall:.1f} s wall, "
                                      f"torch {torch.__version__} CPU eager"}


def separable_shortcut(pot, omega, q0, dt, E0, n, K, W, dev):
    """the structure-exploiting O(D) path is reported SEPARATELY from the dense-state kernel, as
    plain trajectory-steps/s with its own byte model (diagonals of the monodromy blocks only)."""
    from semiclassical_amd import propagators as PR
    G = torch.diag(omega)
    dim = omega.shape[0]
    prop = PR.HermanKlukPropagator(G, G, device=dev, exploit_separability=True)
    prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, generator=torch.Generator().manual_seed(1234))
    slots = torch.zeros((K, 5), dtype=torch.float64, device=dev)
    prop.run(pot, dt, max(W, 1), E0, slots=torch.zeros((max(W, 1), 5), dtype=torch.float64, device=dev))
    prop.synchronize()
    walls = []
    for _ in range(5):                        # a 7 ms loop: take the median of five (an occasional ~50 ms stall of the
        t0 = time.perf_counter()              # idle-to-busy transition otherwise dominates the figure)
        prop.run(pot, dt, K, E0, slots=slots)
        torch.cuda.synchronize(dev)
        walls.append(time.perf_counter() - t0)
    wall = float(np.median(walls))
    assert prop._mono_stale and prop._mono_is_diag, "the diagonal-state kernel did not run"
    prop.profile_step_kernel = True           # kernel duration from a second pass (timing events perturb this short loop)
    prop.run(pot, dt, K, E0, slots=slots)
    prop.profile_step_kernel = False
    kern_ms = float(np.mean(prop.step_kernel_times_ms()))
    nbytes = (12 * dim + 8) * 8 * n          # q, p, 4 diagonals, S, c2, sign: read + write per trajectory step
    return {"value": n * K / wall, "unit": "trajectory-steps/s", "ms_per_step": wall / K * 1e3,
            "kernel": "hk_diag_step_kernel<true>", "kernel_ms": kern_ms, "bytes_per_launch": nbytes,
            "achieved_GBps": nbytes / (kern_ms * 1e-3) / 1e9,
            "note": "opt-in HermanKlukPropagator(exploit_separability=True): diagonal monodromy blocks, c2 = product of the "
                    "diagonal prefactor; NOT the dense-state kernel the roofline object describes"}


def launch_check():
    """--launch-check: every rank joins a gloo group, proves it with one all-reduce and reports its coordinates.
    Exercises the self-launcher without a GPU and without importing the engine (tests/test_distributed.py)."""
    import torch.distributed as dist
    from semiclassical_amd import distributed as D
    rank, world, local = D.init_from_env(backend="gloo")
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t)
    assert "semiclassical_amd._lib" not in sys.modules
    print(json.dumps({"rank": rank, "world": world, "local_rank": local, "sum": float(t.item())}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ntraj", type=int, default=None, help="trajectories per GPU (default 100000 = BASELINE configs[1])")
    ap.add_argument("--ntraj-total", type=int, default=None,
                    help="trajectories of the whole job, sharded over the GPUs (default at --gpus 8: 10^6 = BASELINE configs[3])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-configuration lines (configs object)")
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Self-launch: this parent never touches the GPU; it starts one fresh interpreter per GPU (rank r -> cuda:r,
        # RCCL process group over 127.0.0.1) and exits with their status.  Rank 0 prints the JSON line.
        from semiclassical_amd import distributed as D
        sys.exit(D.launch_local_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    if args.launch_check:
        return launch_check()

    torch.set_default_dtype(torch.float64)
    from semiclassical_amd import distributed as D
    rank, world, local = D.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    from semiclassical_amd import potentials as P, propagators as PR

    if args.ntraj_total is None and args.ntraj is None and world == 8:
        args.ntraj_total = 1000000            # BASELINE.json configs[3]: 10^6 trajectories over 8 GPUs
    if args.ntraj_total is not None:
        n, n_total = D.shard_count(args.ntraj_total, rank, world), args.ntraj_total
    else:
        n = 100000 if args.ntraj is None else args.ntraj
        n_total = n * world
    omega, chi, nac, q0, dt = as60_model()
    dim = omega.shape[0]
    G = torch.diag(omega)
    E0 = float(0.5 * omega.sum())
    pot = P.MorsePotential(omega, chi.clone(), nac)
    prop = PR.HermanKlukPropagator(G, G, device=dev)
    gen = torch.Generator().manual_seed(1234 + rank)
    prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, ntraj_total=n_total, generator=gen)

    K, W = args.steps, args.warmup
    slots = torch.zeros((K, 5), dtype=torch.float64, device=dev)
    wslots = torch.zeros((max(W, 1), 5), dtype=torch.float64, device=dev)
    if W > 0:
        prop.run(pot, dt, W, E0, slots=wslots)
        D.flush_correlations(wslots)
    prop.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- timed region: exactly K steps + the flush ----
    prop.profile_step_kernel = True     # HIP events around the step-kernel launches (same stream)
    barrier()
    t0 = time.perf_counter()
    prop.run(pot, dt, K, E0, slots=slots)
    D.flush_correlations(slots)
    barrier()
    wall = time.perf_counter() - t0
    prop.profile_step_kernel = False
    prop.synchronize()

    tmax = torch.tensor([wall], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall = float(tmax.item())
    cauto, kic = prop.finalize_slots(slots, prop.t - K * dt, dt, E0)
    assert np.isfinite(cauto).all() and np.isfinite(kic).all(), "NaN in correlation functions"

    if rank == 0:
        step_ms = prop.step_kernel_times_ms()
        kern_ms = float(np.mean(step_ms))
        abytes = algorithmic_bytes_per_traj_step(dim) * n
        achieved = abytes / (kern_ms * 1e-3) / 1e9
        traffic, traffic_source = profiled_traffic(n, dim)
        out = {
            "metric": "trajectory-steps/sec + wall-time to converged C(t), anharmonic-AS D=60",
            "value": n_total * K / wall, "unit": "trajectory-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": wall / K * 1e3,
            "wall_time_of_timed_loop_s": wall,       # with --steps 2000: the wall time to the full C(t) of BASELINE configs[1]
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "anharmonic-AS 60-mode, HK, fp64, dt=0.005 fs (BASELINE.json configs[1])",
                       "trajectories_per_gpu": n, "trajectories_total": n_total, "dim": dim,
                       "sharding": f"{world} x {n} trajectories, one RCCL all-reduce of 4*K doubles per flush"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "hk_step_sd_kernel<4,4,true> (+ its hk_modes_kernel pre-pass, same event bracket)",
                         "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": abytes},
            "C_auto_last": [float(cauto[-1].real), float(cauto[-1].imag)],
        }
        if world == 1:
            out["separable_shortcut"] = separable_shortcut(pot, omega, q0, dt, E0, n, K, W, dev)
        if not args.no_cpu_baseline and world == 1:          # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(omega, chi, nac, q0, dt)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
