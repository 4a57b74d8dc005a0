#!/usr/bin/env python
"""Ablation timing of the step kernel on the bench workload (debug tool, GPU box only).

Usage: python tools/phase_timing.py [ntraj]
Times (HIP events on the launch stream): the full step and the prefactor-only launch (loads + matrix + elimination, no
RK4 / stores).  The elimination is ablated at COMPILE time (a run-time switch costs the kernel its register allocation):
    tools/mkvar.sh nolu -DSC_SD_ABLATE_LU=1 ;  SC_LIB_PATH=var/libsc_nolu.so SC_DEBUG_SKIP_LU=1 python tools/phase_timing.py
then reports the streaming phase alone (SC_DEBUG_SKIP_LU only tells sc_hk_step to skip the fix-up launch).
"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from semiclassical_amd import potentials as P, propagators as PR  # noqa: E402
from semiclassical_amd._lib import lib, check, ptr  # noqa: E402

torch.set_default_dtype(torch.float64)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
DIM = int(os.environ.get("DIM", "60"))
omega, chi, nac, q0, dt = bench.as60_model(DIM)
G = torch.diag(omega)
pot = P.MorsePotential(omega, chi.clone(), nac)
prop = PR.HermanKlukPropagator(G, G, device="cuda")
prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, generator=torch.Generator().manual_seed(1))


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


desc = prop._potential_descriptor(pot)
if os.environ.get("LAYOUT", "rowmajor") == "tiled":          # storage order of the monodromy blocks (SC_MONO_TILED16)
    from semiclassical_amd import _lib
    prop._set_mono_layout(_lib.SC_MONO_TILED16)
print("mono_layout", prop._state.mono_layout, "tuning build", lib.sc_tuning_build())
full = lambda: check(lib.sc_hk_step(desc, prop._state, prop._hk, dt, 0, ptr(prop._epart), prop._stream()))
pref = lambda: check(lib.sc_hk_step(desc, prop._state, prop._hk, dt, 1, None, prop._stream()))
ab = bench.algorithmic_bytes_per_traj_step(DIM) * n
for occ in os.environ.get("OCC_LIST", "2").split(","):
    os.environ["SC_SD_OCC"] = occ
    t_full = timed(full)
    t_pref = timed(pref)
    what = "WITHOUT elimination" if os.environ.get("SC_DEBUG_SKIP_LU") else "full"
    print(f"D={DIM} n={n} occ={occ} [{what}]: step {t_full:.3f} ms ({ab / t_full / 1e6:.0f} GB/s algorithmic) | "
          f"prefactor-only launch (no RK4 / stores) {t_pref:.3f} ms", flush=True)
if prop._state.mono_layout == 1 and lib.sc_hk_step_multi_supported(desc, prop._state, prop._hk):
    prop._launch_step_pair(desc, dt)              # allocates the scratch of sc_hk_step_multi
    m = prop._multi
    pair = lambda: check(lib.sc_hk_step_multi(desc, prop._state, prop._hk, m["ms"], dt, ptr(m["epart"]), prop._stream()))
    t_pair = timed(pair)
    print(f"D={DIM} n={n} two steps per visit [{what}]: {t_pair / 2:.3f} ms per step ({t_pair:.3f} per launch)", flush=True)
