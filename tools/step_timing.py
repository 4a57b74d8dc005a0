#!/usr/bin/env python
"""Duration of the fast-path step launch (sc_hk_step, mode 0) on the bench workload (GPU box; A/B tool for variant
libraries: SC_LIB_PATH=var/libsc_X.so python tools/step_timing.py [ntraj])."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from semiclassical_amd import _lib, potentials as P, propagators as PR  # noqa: E402
from semiclassical_amd._lib import lib, check, ptr  # noqa: E402

torch.set_default_dtype(torch.float64)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
omega, chi, nac, q0, dt = bench.as60_model(int(os.environ.get("DIM", "60")))
G = torch.diag(omega)
pot = P.MorsePotential(omega, chi.clone(), nac)
prop = PR.HermanKlukPropagator(G, G, device="cuda")
prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, generator=torch.Generator().manual_seed(1))
desc = prop._potential_descriptor(pot)
if os.environ.get("LAYOUT", "tiled") == "tiled":
    prop._set_mono_layout(_lib.SC_MONO_TILED16)
full = lambda: check(lib.sc_hk_step(desc, prop._state, prop._hk, dt, 0, ptr(prop._epart), prop._stream()))
for _ in range(3):
    full()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    full()
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10
ab = bench.algorithmic_bytes_per_traj_step(omega.shape[0]) * n
print(f"{os.path.basename(os.environ.get('SC_LIB_PATH', 'product'))} n={n}: step launch {t:.3f} ms  ({ab / t / 1e6:.0f} GB/s algorithmic)  "
      f"c2[0]={prop._c2[0].item():.6f}", flush=True)
