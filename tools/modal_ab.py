#!/usr/bin/env python
"""run() of the harmonic methylium example (HK, n = 1e5): product with Phi against normal-mode coordinates (sc_hk_run_modal)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tests import cases
from tests.engine_cases import engine_potential
from semiclassical_amd import propagators as PR
torch.set_default_dtype(torch.float64)
g = cases.load("hk_methylium")
pot = engine_potential(g)
G0 = cases.T(g["Gamma_0"])
q0 = cases.T(g["q0"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dt, E0 = float(g["dt"]), float(g["E0"])
out = []
for frm in (16, 10 ** 9):
    prop = PR.HermanKlukPropagator(G0, G0, device="cuda")
    prop.normal_modes_from = frm
    prop.initial_conditions(q0, 0.0 * q0, G0, ntraj=n, seed=5)
    prop.run(pot, dt, 20, E0)
    prop.synchronize()
    t0 = time.perf_counter()
    c, k = prop.run(pot, dt, nt, E0)
    prop.synchronize()
    wall = time.perf_counter() - t0
    print("normal modes" if frm == 16 else "product with Phi", "modal cache:", bool(prop.__dict__.get("_modal_cache")), f"{wall / nt * 1e3:.4f} ms per step")
    out.append((c, k, prop.y.clone(), prop._c2.clone(), prop._sgn.clone()))
a, b = out
rel = lambda x, y: float(np.max(np.abs(x - y)) / np.max(np.abs(y)))
print("C", rel(a[0], b[0]), "k", rel(a[1], b[1]), "y", rel(a[2].cpu().numpy(), b[2].cpu().numpy()),
      "c2", rel(a[3].cpu().numpy(), b[3].cpu().numpy()), "signs equal", torch.equal(a[4], b[4]))
