#!/usr/bin/env python
"""Headline benchmark: trajectory-steps/s of the HK loop on the synthetic 60-mode anharmonic-AS model.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--ntraj n_per_gpu | --ntraj-total n]

A "step" is one pass of the hot path over one batch: (C_auto, k_ic, RK4 step + prefactor) for all
trajectories (the loop body of reference cli.py:401-436).  Workload = BASELINE.json configs[1]:
anharmonic-AS, D = 60, 10^5 trajectories per GPU, HK, fp64, dt = 0.005 fs.
Inputs are resident in HBM before the timed region; the timed region ends with the single all-reduce
of the accumulated correlation sums (the "flush").  One JSON line is printed by rank 0.

--gpus N without WORLD_SIZE in the environment: this process only starts N rank processes (one per GPU, RCCL
process group) and waits; it never touches the GPU itself.  At --gpus 8 the default size is BASELINE configs[3]:
10^6 trajectories in total = 8 x 125 000.

At N = 1 the line also carries: `configs` (the other single-GPU BASELINE configurations with their dominant kernels'
durations and roofline fractions), `wall_to_full_Ct_s` (a real 2000-step run of the headline configuration),
`separable_shortcut` (the opt-in structure-exploiting path, reported apart) and `cpu_baseline`.
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
GOLDEN = os.path.join(ROOT, "tests", "golden")

HBM_PEAK_GBS = 8000.0     # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling
FP64_PEAK_TFLOPS = 78.6   # MI355X datasheet, vector = matrix FP64 (the guide gives no FP64 figure)
# what the two FP64 pipes SUSTAIN on this pool's boxes (micro-benchmarks, profiles/): back-to-back v_fmac_f64 on 16 independent
# accumulators, two waves per SIMD (r3_dpp_fmac.txt: 57-61 TFLOP/s), back-to-back v_mfma_f64_16x16x4_f64 (r4_mfma_f64.txt: 77)
FP64_VALU_SUSTAINED_TFLOPS = 59.0
FP64_MFMA_SUSTAINED_TFLOPS = 77.0


def as60_model(dim=60):
    """synthetic 60-mode AS model, SURVEY.md section 8d config 2 (other `dim` only for the tools/ experiments)"""
    from semiclassical_amd.synthetic import anharmonic_as_model
    return anharmonic_as_model(dim)


def algorithmic_bytes_per_traj_step(D):
    """read y + write y + per-trajectory side data = 64 D^2 + 64 D + 72"""
    return 64 * D * D + 64 * D + 72


def wm_flops_per_traj_step(D, dp):
    """real flops (2 per multiply-add) of the Walton-Manolopoulos prefactor + correlation terms as the engine computes
    them (csrc/sc_wm_small.hip header; docs/NOTEBOOK.md section 4.2a): Mq', Mp', Gt Mq', the two e x e Gram matrices, the
    e x e complex elimination with D right-hand sides, eqns (57), (59), (70), the projected M', its elimination"""
    E = 2 * dp
    fma = (2 * D * D * E + D * D * E + 2 * E * E * D            # Mq', Mp' ; Tq ; G, S
           + 4 * E * E * (E // 2 + D)                           # Gauss-Jordan on A' with D right-hand sides (complex)
           + 4 * D * D * E + 2 * D * D * E                      # (57), (59)
           + 2 * D ** 3 + 4 * D ** 3                            # V, CQQ (70)
           + 2 * D * D * dp + 2 * dp * dp * D + 10 * D * dp     # M', hat vectors
           + 4 * dp * dp * (dp // 2 + 5))                       # Gauss-Jordan on M' with 5 right-hand sides
    return 2 * fma


def profiled_traffic(n, dim, live_ms, steps_per_launch=1):
    """HBM bytes per step-kernel launch from the newest committed rocprofv3 PMC passes (profiles/rN_hbm_traffic.json) --
    but only if that profile still describes THIS run: same workload, same kernel, and the kernel-trace summary of the same
    profiling session (profiles/rN_bench_kernel_stats.csv) gives the launch pair (step kernel + modes pre-pass) within 5 % of
    the duration measured live (``live_ms``).  Otherwise (None, reason): a stale constant must not pose as a measurement."""
    import csv
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")),
                   key=lambda f: int(re.match(r"r(\d+)_", os.path.basename(f)).group(1)), reverse=True)
    reasons = []
    for path in files:
        name = os.path.basename(path)
        tag = name.split("_")[0]
        try:
            with open(path) as f:
                t = json.load(f)
            if t["workload"]["ntraj"] != n or t["workload"]["dim"] != dim:
                continue
            kernel = t["workload"]["kernel"]
            per = int(t.get("steps_per_launch", 1))
            want = f"hk_step_sd_kernel<4, 4, true, true, {steps_per_launch}>"
            if want not in kernel or per != steps_per_launch:
                reasons.append(f"profiles/{name} profiled {kernel}, not the kernel of this run ({want})")
                continue
            stats = os.path.join(ROOT, "profiles", f"{tag}_bench_kernel_stats.csv")
            avg = {}
            with open(stats) as f:
                for row in csv.reader(f):
                    if len(row) >= 4 and row[0] != "Name" and not row[0].startswith("#"):
                        avg[row[0]] = float(row[3]) / 1e3
            step = next(v for k, v in avg.items() if want in k) / per
            modes = next(v for k, v in avg.items() if ("hk_modes_multi_kernel" if per > 1 else "hk_modes_kernel") in k) / per
            if abs(step + modes - live_ms) > 0.05 * live_ms:
                reasons.append(f"profiles/{name}: its session measured {step + modes:.3f} ms per time step, this run {live_ms:.3f} ms "
                          "(more than 5 % apart: the profile is stale for this build or box)")
                continue
            return t["traffic_bytes_per_launch"], (f"profiles/{name} (FETCH_SIZE calibrated + WRITE_SIZE, separate passes; kernel-trace of "
                                                   f"that session: {step + modes:.3f} ms per time step, this run {live_ms:.3f} ms)")
        except (OSError, KeyError, ValueError, StopIteration) as err:
            reasons.append(f"profiles/{name}: {type(err).__name__} {err}")
            continue
    return None, (reasons[0] if reasons else "no profiles/rN_hbm_traffic.json for this workload")


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


# ----------------------------------------------------------------------------------------------------------------------
# the other single-GPU configurations of BASELINE.json (driver-timed, N = 1 only)
# ----------------------------------------------------------------------------------------------------------------------

def _load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def _T(x):
    return torch.from_numpy(np.asarray(x)).clone()


def _timed_loop(prop, pot, dt, E0, steps, dev, use_graph=False, reps=3):
    """median wall time of `reps` fused loops of `steps` steps (after one warm-up loop); seconds"""
    slots = torch.zeros((steps, 5), dtype=torch.float64, device=dev)
    prop.run(pot, dt, min(steps, 3), E0, slots=slots, use_graph=False)
    torch.cuda.synchronize(dev)
    walls = []
    for _ in range(reps):
        t0 = time.perf_counter()
        prop.run(pot, dt, steps, E0, slots=slots, use_graph=use_graph)
        torch.cuda.synchronize(dev)
        walls.append(time.perf_counter() - t0)
    prop.synchronize()
    return float(np.median(walls))


def _kernel_ms(prop, pot, dt, E0, steps, dev):
    """mean duration per launch of every labelled kernel over one more loop (HIP events on the launch stream)"""
    prop.__dict__.pop("_kernel_events", None)
    prop.kernel_timing = True
    prop.run(pot, dt, steps, E0, slots=torch.zeros((steps, 5), dtype=torch.float64, device=dev))
    prop.kernel_timing = False
    return {k: float(np.mean(v)) for k, v in prop.kernel_times_ms().items()}


def config1(dev, n, steps):
    """configs[0]: 5-mode anharmonic AS (the reference's tests/DATA/AnharmonicAS/5modes model, chi = 0.02), HK"""
    from semiclassical_amd import potentials as P, propagators as PR
    g = _load("hk_as5_chi002")
    pot = P.MorsePotential(_T(g["omega"]), _T(g["chi"]), _T(g["nac"]))
    G = _T(g["Gamma_i"])
    prop = PR.HermanKlukPropagator(G, G, device=dev)
    prop.initial_conditions(_T(g["q0"]), _T(g["p0"]), _T(g["Gamma_0"]), ntraj=n, generator=torch.Generator().manual_seed(7))
    dt, E0, D = float(g["dt"]), float(g["E0"]), 5
    # run(): separable potential, diagonal widths, D <= 12 -> the whole loop is ONE launch (sc_hk_run)
    wall = _timed_loop(prop, pot, dt, E0, steps, dev)
    nbytes = algorithmic_bytes_per_traj_step(D) * n
    out = {"workload": f"anharmonic-AS 5-mode, HK, n={n} (BASELINE.json configs[0])", "n": n, "steps": steps,
           "ms_per_step": wall / steps * 1e3, "value": n * steps / wall, "unit": "trajectory-steps/s",
           "kernel": "hk_run_sep16_kernel<8,MORSE>: the caller loop (C_auto, k_ic, step) x steps in one launch, state in registers",
           "equivalent_stepwise_GBps": nbytes / (wall / steps) / 1e9,
           "equivalent_stepwise_note": "algorithmic bytes of a step-at-a-time engine (SURVEY 8d) over the time per step of the "
                                       "whole-loop launch -- NOT achieved bandwidth: the launch reads and writes a trajectory once per "
                                       "run(); what bounds it is FP64 VALU issue and latency (profiles/r3_config1_pmc.json)"}
    # the same loop step by step (six launches per step), eager and replayed from a HIP graph
    prop._whole_loop_ok = False
    wall_s = _timed_loop(prop, pot, dt, E0, steps, dev)
    out["stepwise"] = {"ms_per_step": wall_s / steps * 1e3, "value": n * steps / wall_s}
    if n <= 10000:
        wall_g = _timed_loop(prop, pot, dt, E0, steps, dev, use_graph=True)
        out["stepwise"].update({"ms_per_step_graph": wall_g / steps * 1e3,
                                "note": "eager: ~6 ctypes launches per step from Python; graph: one hipGraphLaunch per step"})
    k = _kernel_ms(prop, pot, dt, E0, steps, dev)
    out["stepwise"].update({"kernel": "hk_step_sep16_kernel<8,true,MORSE> (four trajectories per wavefront)", "kernel_ms": k.get("hk_step"),
                            "hbm_GBps": nbytes / (k["hk_step"] * 1e-3) / 1e9})
    return out


def config3(dev, n, steps):
    """configs[2]: harmonic methylium (12 Cartesian coordinates, rank-6 Gamma_0), WM with the default cell width 1e4"""
    from semiclassical_amd import potentials as P, propagators as PR
    g = _load("wm_methylium")
    pot = P.MolecularHarmonicPotential.from_arrays(g["pos0"], g["energy0"], g["grad0"], g["hess0"], g["masses"], g["nac0"],
                                                   origin=float(g["origin"]))
    Gi = _T(g["Gamma_i"])
    prop = PR.WaltonManolopoulosPropagator(Gi, Gi, float(g["alpha"]), float(g["beta"]), device=dev)
    prop.initial_conditions(_T(g["q0"]), _T(g["p0"]), _T(g["Gamma_0"]), ntraj=n, generator=torch.Generator().manual_seed(7))
    dt, E0, D, dp = float(g["dt"]), float(g["E0"]), 12, 6
    wall = _timed_loop(prop, pot, dt, E0, steps, dev)
    k = _kernel_ms(prop, pot, dt, E0, steps, dev)
    flops = wm_flops_per_traj_step(D, dp) * n
    nbytes = (4 * D * D + 4 * D + 16) * 8 * n            # blocks, q, p, z_i, scalars and trackers of a trajectory
    wm_ms, hk_ms = k["wm"], k["hk_step"]
    return {"workload": f"harmonic methylium D=12 d'=6, WM alpha=beta=1e4, n={n} (BASELINE.json configs[2])", "n": n, "steps": steps,
            "ms_per_step": wall / steps * 1e3, "value": n * steps / wall, "unit": "trajectory-steps/s",
            "kernels_ms": {"wm_small_kernel<12,6>": wm_ms, "hk_step_lin_kernel<12,6,false> (RK4 + HK prefactor)": hk_ms},
            "roofline": {"kernel": "wm_small_kernel<12,6>", "bound": "fp64", "achieved": flops / (wm_ms * 1e-3) / 1e12,
                         "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / (wm_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                         "measured_pipe_rate": FP64_VALU_SUSTAINED_TFLOPS,
                         "frac_of_measured_pipe_rate": flops / (wm_ms * 1e-3) / 1e12 / FP64_VALU_SUSTAINED_TFLOPS,
                         "algorithmic_flops_per_launch": flops,
                         "hbm_GBps": nbytes / (wm_ms * 1e-3) / 1e9, "hbm_frac": nbytes / (wm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "hk_step_roofline": {"bound": "hbm", "achieved": algorithmic_bytes_per_traj_step(D) * n / (hk_ms * 1e-3) / 1e9,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": algorithmic_bytes_per_traj_step(D) * n / (hk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}


def config3_hk(dev, n, steps):
    """the same methylium model with the HK propagator (the reference's own methylium_AH example runs both): the constant-Hessian
    step kernel plus the correlation kernel for DENSE, rank-deficient width matrices"""
    from semiclassical_amd import potentials as P, propagators as PR
    g = _load("hk_methylium")
    pot = P.MolecularHarmonicPotential.from_arrays(g["pos0"], g["energy0"], g["grad0"], g["hess0"], g["masses"], g["nac0"],
                                                   origin=float(g["origin"]))
    Gi = _T(g["Gamma_i"])
    prop = PR.HermanKlukPropagator(Gi, Gi, device=dev)
    prop.initial_conditions(_T(g["q0"]), _T(g["p0"]), _T(g["Gamma_0"]), ntraj=n, generator=torch.Generator().manual_seed(7))
    dt, E0 = float(g["dt"]), float(g["E0"])
    wall = _timed_loop(prop, pot, dt, E0, steps, dev)
    k = _kernel_ms(prop, pot, dt, E0, steps, dev)
    return {"workload": f"harmonic methylium D=12 d'=6, HK, n={n} (not a BASELINE configuration: the HK half of the reference's methylium example)",
            "n": n, "steps": steps, "ms_per_step": wall / steps * 1e3, "value": n * steps / wall, "unit": "trajectory-steps/s",
            "kernels_ms": {kk: vv for kk, vv in k.items()}}


def config5(dev, n, steps):
    """configs[4]: sGDML potential, 30 atoms (synthetic model of SURVEY.md section 8d: D = 90, Dd = 435, M = 200), HK"""
    from semiclassical_amd import propagators as PR
    from semiclassical_amd.gdml import MolecularGDMLPotential
    from semiclassical_amd.synthetic import sgdml_model, ArrayFchk
    N, M = 30, 200
    model, pos = sgdml_model(N, M, 30)
    pot0 = MolecularGDMLPotential(model, ArrayFchk(np.ones(3 * N), np.zeros(3 * N), model["z"]))
    _, g0, _ = pot0.harmonic_approximation(torch.from_numpy(pos.reshape(-1, 1)).to(dev))
    model["R_d_desc_alpha"] = model["R_d_desc_alpha"] * (0.02 / float(g0.abs().max()))        # molecular-size forces
    masses = np.repeat(np.full(N, 12.0 * 1822.888), 3)
    pot = MolecularGDMLPotential(model, ArrayFchk(masses, np.zeros(3 * N), model["z"]))
    q0 = torch.from_numpy(pos.reshape(-1))
    G = torch.diag(torch.full((3 * N,), 40.0))
    prop = PR.HermanKlukPropagator(G, G, device=dev)
    prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, generator=torch.Generator().manual_seed(7))
    D, Dd = 3 * N, N * (N - 1) // 2
    wall = _timed_loop(prop, pot, 2.0, 0.0, steps, dev, reps=3)
    k = _kernel_ms(prop, pot, 2.0, 0.0, steps, dev)
    # the two kernels inside the dense_mono_step bracket: the prefactor alone (mode 1 of the same entry point) in its own
    # bracket, the MFMA RK4 kernel as the difference (profiles/*_config5_kernel_stats.csv holds both from rocprofv3)
    from semiclassical_amd._lib import lib, check
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for e0, e1 in ev:
        e0.record()
        check(lib.sc_dense_mono_step(prop._state, prop._hk, None, None, prop._mono_sums_ptr(), 0.0, 1, prop._stream()))
        e1.record()
    torch.cuda.synchronize(dev)
    pref_ms = float(np.mean([a.elapsed_time(b) for a, b in ev[1:]]))
    # sGDML evaluation: three rank-M sums over (3N)^2 (6 M (3N)^2 flops), J^T products 2 x 2 M Dd 3 ... ; monodromy RK4 16 D^3
    stage_flops = (6 * M * D * D + 8 * M * Dd * 3 + 4 * M * Dd) * n
    mono_flops = 16 * D ** 3 * n
    return {"workload": f"sGDML 30-atom synthetic model (D=90, M=200), HK, n={n} (BASELINE.json configs[4]: 10^4 over 8 GPUs = 1250 per GPU)",
            "n": n, "steps": steps, "ms_per_step": wall / steps * 1e3, "value": n * steps / wall, "unit": "trajectory-steps/s",
            "kernels_ms": {"gdml_stage_kernel (x4 per step)": k["gdml_stage"], "dense_mono_step (MFMA RK4 + prefactor)": k["dense_mono_step"],
                           "dense_prefactor_reg_kernel (own bracket)": pref_ms,
                           "dense_mono_mfma_slab_kernel (difference)": k["dense_mono_step"] - pref_ms},
            "roofline": {"kernel": "gdml_stage_kernel", "bound": "fp64", "achieved": stage_flops / (k["gdml_stage"] * 1e-3) / 1e12,
                         "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": stage_flops / (k["gdml_stage"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                         "measured_pipe_rate": FP64_VALU_SUSTAINED_TFLOPS,
                         "frac_of_measured_pipe_rate": stage_flops / (k["gdml_stage"] * 1e-3) / 1e12 / FP64_VALU_SUSTAINED_TFLOPS,
                         "algorithmic_flops_per_launch": stage_flops},
            "dense_mono_roofline": {"kernel": "dense_mono_mfma_slab_kernel", "bound": "fp64 (MFMA)",
                                    "achieved": mono_flops / ((k["dense_mono_step"] - pref_ms) * 1e-3) / 1e12,
                                    "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": mono_flops / ((k["dense_mono_step"] - pref_ms) * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                                    "measured_pipe_rate": FP64_MFMA_SUSTAINED_TFLOPS,
                                    "frac_of_measured_pipe_rate": mono_flops / ((k["dense_mono_step"] - pref_ms) * 1e-3) / 1e12
                                                                  / FP64_MFMA_SUSTAINED_TFLOPS}}


def other_configs(dev):
    out = {}
    for key, fn, args in (("config1_n1000", config1, (1000, 100)), ("config1_n100000", config1, (100000, 50)),
                          ("config3_wm_methylium", config3, (100000, 30)), ("methylium_hk", config3_hk, (100000, 200)),
                          ("config5_gdml30_share", config5, (1250, 5)), ("config5_gdml30_n10000", config5, (10000, 3))):
        try:
            out[key] = fn(dev, *args)
        except Exception as err:      # the headline line is still printed -- and the process then exits non-zero (main)
            import traceback
            traceback.print_exc(file=sys.stderr)
            out[key] = {"error": f"{type(err).__name__}: {err}"}
        torch.cuda.empty_cache()
    return out


def wall_to_full_ct(pot, omega, q0, dt, E0, n, dev, nt=2000):
    """the second half of the metric: wall time of the full correlation function (nt = 2000 steps of 0.005 fs,
    README.rst:324-327) of the headline configuration, initial conditions included"""
    from semiclassical_amd import propagators as PR
    G = torch.diag(omega)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    prop = PR.HermanKlukPropagator(G, G, device=dev)
    prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, seed=99)          # sampled on the device
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    cauto, kic = prop.run(pot, dt, nt, E0)
    wall = time.perf_counter() - t0
    assert np.isfinite(cauto).all() and np.isfinite(kic).all() and abs(cauto[0] - 1.0) < 1e-3
    # How converged is that C(t)?  The reference has no criterion (its only diagnostic is the O(n^2) norm, cli.py:424-429);
    # C(t) is a Monte-Carlo mean, so its statistical error is the yardstick: with c_i the weighted per-trajectory terms
    # (C = sum_i c_i), the standard error of the sum is sqrt(N sum_i |c_i|^2 - |C|^2) / sqrt(N - 1)  (outside the timed region).
    prop._correlate_current(False)
    ci = prop._cq
    total = torch.sum(ci)
    se = float(torch.sqrt(torch.clamp(n * torch.sum(torch.abs(ci) ** 2) - torch.abs(total) ** 2, min=0.0) / (n - 1)).item())
    return {"value": wall, "unit": "s", "steps": nt, "trajectories": n, "initial_conditions_s": t1 - t0,
            "loop_s": wall - (t1 - t0), "C_auto_last": [float(cauto[-1].real), float(cauto[-1].imag)],
            "definition": f"wall time of the full correlation function: {nt} steps of 0.005 fs x {n} trajectories, initial conditions "
                          "included; 'converged' is quantified by the Monte-Carlo standard error below, not by a stopping rule",
            "mc_standard_error_of_C_at_last_step": se,
            "trajectories_for_1e-3_standard_error": int(np.ceil(n * (se / 1e-3) ** 2)),
            "wall_to_1e-3_standard_error_s_extrapolated": (wall - (t1 - t0)) * (se / 1e-3) ** 2}


# ----------------------------------------------------------------------------------------------------------------------

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
    # one write per line: the ranks share the parent's stdout, print() would emit text and newline separately
    sys.stdout.write(json.dumps({"rank": rank, "world": world, "local_rank": local, "sum": float(t.item())}) + "\n")
    sys.stdout.flush()
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
    ap.add_argument("--no-configs", action="store_true", help="skip the per-configuration lines and the 2000-step run")
    ap.add_argument("--no-pairs", action="store_true", help="A/B: one step-kernel launch per time step instead of two steps per visit")
    ap.add_argument("--share-gpu", action="store_true",
                    help="REHEARSAL of the multi-rank flow on a one-GPU box: every rank runs on cuda:0, process group over gloo; the line is "
                         "marked and its numbers mean nothing")
    ap.add_argument("--config", choices=["1", "3", "3hk", "5"], default=None,
                    help="run ONLY that BASELINE configuration's side measurement (for one rocprofv3 summary per configuration: "
                         "1 = 5-mode AS at n = 1e5, 3 = methylium WM at n = 1e5, 5 = 30-atom sGDML at n = 1e4) and print its JSON")
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
    if args.config is not None:
        dev = torch.device("cuda", 0)
        fn, fargs = {"1": (config1, (100000, 50)), "3": (config3, (100000, 30)), "3hk": (config3_hk, (100000, 200)), "5": (config5, (10000, 3))}[args.config]
        print(json.dumps({f"config{args.config}": fn(dev, *fargs)}), flush=True)
        return
    from semiclassical_amd import distributed as D
    if args.share_gpu:
        os.environ["SC_DIST_BACKEND"] = "gloo"
    rank, world, local = D.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.share_gpu:
        local = 0
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
    # initial conditions are sampled on the device (sc_sample_initial): every rank draws ITS slice of one global
    # ensemble -- deviate j of global trajectory i depends on (seed, i, j) only, not on the number of ranks
    first = D.shard_slice(n_total, rank, world).start if args.ntraj_total is not None else rank * n
    prop.pair_steps = not args.no_pairs
    prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, ntraj_total=n_total, seed=1234, first_index=first)

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
    # who took part: backend of the process group and the device every rank ran on (so that a scaling record shows that
    # RCCL -- torch's "nccl" backend on ROCm -- saw N ranks on N different GPUs)
    me = {"rank": rank, "device": str(dev), "gpu": torch.cuda.get_device_name(dev), "trajectories": n,
          "pci_bus_id": getattr(torch.cuda.get_device_properties(dev), "pci_bus_id", None)}
    ranks = [me]
    backend = "none (single process)"
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
        backend = dist.get_backend()

    if rank == 0:
        step_ms = prop.step_kernel_times_ms()
        kern_ms = float(np.mean(step_ms))
        abytes = algorithmic_bytes_per_traj_step(dim) * n
        achieved = abytes / (kern_ms * 1e-3) / 1e9
        per = 2 if (prop._multi is not None and K >= 2) else 1        # run() advanced two time steps per launch (sc_hk_step_multi)
        traffic, traffic_source = profiled_traffic(n, dim, kern_ms, per)
        out = {
            "metric": "trajectory-steps/sec + wall-time to converged C(t), anharmonic-AS D=60",
            "value": n_total * K / wall, "unit": "trajectory-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": wall / K * 1e3,
            "wall_time_of_timed_loop_s": wall,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "scaling_note": (f"{n} trajectories per GPU at N={world}"
                             + ("; N=8 runs BASELINE configs[3] as stated (10^6 = 8 x 125000) while N<8 run 100000 per GPU "
                                "(configs[1]): per-GPU work differs by 25 %, the metric is a rate and the kernel's rate does not "
                                "depend on n at this size" if (n_total == 1000000 and world == 8) else "")),
            "backend": backend, "ranks": ranks,
            "dtype": "f64", "data": "synthetic" if not args.share_gpu else "synthetic; REHEARSAL (--share-gpu): all ranks on one GPU, not a measurement",
            "config": {"workload": "anharmonic-AS 60-mode, HK, fp64, dt=0.005 fs (BASELINE.json configs[1]"
                                   + ("; 10^6 trajectories over 8 GPUs = configs[3])" if n_total == 1000000 and world == 8 else ")"),
                       "trajectories_per_gpu": n, "trajectories_total": n_total, "dim": dim,
                       "sharding": f"{world} x {n} trajectories, one RCCL all-reduce of 4*K doubles per flush"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": (f"hk_step_sd_kernel<4,4,true,true,{per}> (+ its modes pre-pass, same event bracket)"
                                    + ("; one launch advances every trajectory by TWO time steps (the second step's reads come from "
                                       "the memory-side cache): achieved, traffic, kernel_ms and the algorithmic bytes are per TIME STEP, "
                                       "i.e. per half launch" if per == 2 else "")),
                         "steps_per_launch": per,
                         "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": abytes,
                         "also_bound_by": ("the pivot hand-over chain of the 60 x 60 complex elimination (latency, not bytes): the streaming phase "
                                           "alone runs at 2.8 ms per step in this mode (profiles/r4_sd_phases.txt, r4_lu_pivot_clock.txt) "
                                           if per == 2 else None)},
            "C_auto_last": [float(cauto[-1].real), float(cauto[-1].imag)],
        }
        if world == 1:
            del prop
            torch.cuda.empty_cache()
            out["separable_shortcut"] = separable_shortcut(pot, omega, q0, dt, E0, n, K, W, dev)
            if not args.no_configs:
                # no empty_cache() here: the state buffers of the run above are reused (returning 12 GB to the driver and
                # asking for them again costs 1.8 s of hipFree / hipMalloc inside the timed region)
                out["wall_to_full_Ct_s"] = wall_to_full_ct(pot, omega, q0, dt, E0, n, dev)
                torch.cuda.empty_cache()
                out["configs"] = other_configs(dev)
            if not args.no_cpu_baseline:          # rank 0 at N = 1 only
                out["cpu_baseline"] = cpu_baseline(omega, chi, nac, q0, dt)
        print(json.dumps(out), flush=True)
        broken = [k for k, v in out.get("configs", {}).items() if "error" in v]
        if broken:                                # the line above is complete; a broken side configuration still fails the run
            sys.stderr.write(f"bench.py: side configuration(s) failed: {', '.join(broken)}\n")
            sys.exit(3)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
