#!/usr/bin/env python
"""Achieved deviations of the sGDML path (GPU box): HIP kernel vs the reference's fp64 output, and both vs the
extended-precision evaluation of the same formulas (tests/golden/gdml_coumarin_truth.npz, oracle/gdml_truth.py)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases  # noqa: E402
from tests.engine_cases import engine_potential, engine_propagator  # noqa: E402

torch.set_default_dtype(torch.float64)
cnp = lambda t: t.detach().cpu().numpy()
g, tr = cases.load("gdml_coumarin_eval"), cases.load("gdml_coumarin_truth")
pot = engine_potential(dict(potential="gdml", nac0=np.zeros(51), masses=np.ones(51), origin=0.0))
v, grad, hess = pot.harmonic_approximation(torch.from_numpy(g["r"]).t().contiguous().cuda())
v, grad, hess = cnp(v), cnp(grad.t()), cnp(hess.permute(2, 0, 1))
print("function level (max-norm relative deviations)")
print(f"  HIP vs reference : E {cases.rel_err(v, g['energy']):.2e}  grad {cases.rel_err(grad, g['grad']):.2e}  hess {cases.rel_err(hess, g['hess']):.2e}")
print(f"  HIP vs truth     : E {cases.rel_err(v[:3], tr['energy']):.2e}  grad {cases.rel_err(grad[:3], tr['grad']):.2e}  hess {cases.rel_err(hess[:3], tr['hess']):.2e}")
print(f"  reference vs truth: E {tr['ref_dev'][0]:.2e}  grad {tr['ref_dev'][1]:.2e}  hess {tr['ref_dev'][2]:.2e}")

g = cases.load("hk_coumarin_gdml")
pot, prop = engine_potential(g), engine_propagator(g)
nt, dt, E0 = int(g["nt"]), float(g["dt"]), float(g["E0"])
cauto, kic, dc2, dy = np.zeros(nt, dtype=complex), np.zeros(nt, dtype=complex), 0.0, 0.0
for t in range(nt):
    dc2 = max(dc2, cases.rel_err(cnp(prop._c2), g["c2"][t]))
    cauto[t], kic[t] = prop.autocorrelation(E0), prop.ic_correlation(pot, E0)
    prop.step(pot, dt)
    if t + 1 in g["snaps"]:
        y, d = cnp(prop.y), prop.dim
        dy = max(dy, cases.rel_err(np.vstack((y[:2 * d], y[-1:])), g[f"qpS_{t + 1}"]), cases.rel_err(y[:, 0], g[f"ytraj0_{t + 1}"]))
print(f"HK propagation on the coumarin surface, {nt} steps: c2 {dc2:.2e}  y {dy:.2e}  C(t) {cases.rel_err(cauto, g['cauto']):.2e}  "
      f"k_ic(t) {cases.rel_err(kic, g['kic']):.2e}")
