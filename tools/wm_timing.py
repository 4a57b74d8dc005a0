#!/usr/bin/env python
"""Time of the WM per-trajectory kernel (sc_wm_correlate) alone on the methylium WM case (GPU box; A/B tool)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases  # noqa: E402
from tests.engine_cases import engine_potential  # noqa: E402
from semiclassical_amd import propagators as PR  # noqa: E402

torch.set_default_dtype(torch.float64)
name = os.environ.get("CASE", "wm_methylium")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
g = cases.load(name)
pot = engine_potential(g)
Gi, Gt = cases.T(g["Gamma_i"]), cases.T(g["Gamma_t"])
prop = PR.WaltonManolopoulosPropagator(Gi, Gt, float(g["alpha"]), float(g["beta"]), device="cuda")
prop.initial_conditions(cases.T(g["q0"]), cases.T(g["p0"]), cases.T(g["Gamma_0"]), ntraj=n, generator=torch.Generator().manual_seed(7))
prop.step(pot, float(g["dt"]))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    prop._wm_launch(0)
e1.record()
torch.cuda.synchronize()
print(f"{name} D={prop.dim} n={n}: wm kernel {e0.elapsed_time(e1) / 5:.3f} ms", flush=True)
