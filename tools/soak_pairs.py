#!/usr/bin/env python
"""One-off soak: the pair path against one launch per step over many steps at the benchmark size, bit for bit (the store-data hazard of
DESIGN section 3 showed up in 1 % of the trajectories per step before it was guarded: any residue would break the identity here)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from semiclassical_amd import potentials as P, propagators as PR
torch.set_default_dtype(torch.float64)
n, nt = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, int(sys.argv[2]) if len(sys.argv) > 2 else 200
omega, chi, nac, q0, dt = bench.as60_model(60)
G = torch.diag(omega)
E0 = float(0.5 * omega.sum())
pot = P.MorsePotential(omega, chi.clone(), nac)
out = []
for pairs in (True, False):
    prop = PR.HermanKlukPropagator(G, G, device="cuda")
    prop.pair_steps = pairs
    prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, seed=7)
    c, k = prop.run(pot, dt, nt, E0)
    prop.synchronize()
    out.append((c, k, prop._mono.clone(), prop._c2.clone(), prop._sgn.clone(), prop._qp.clone()))
    del prop
a, b = out
print("C equal", np.array_equal(a[0], b[0]), "k equal", np.array_equal(a[1], b[1]), "mono equal", torch.equal(a[2], b[2]),
      "c2 equal", torch.equal(torch.view_as_real(a[3]), torch.view_as_real(b[3])), "sgn equal", torch.equal(a[4], b[4]), "qp equal", torch.equal(a[5], b[5]))
print("sign flips:", int((a[4] < 0).sum()), "of", n, "| |C(t_end)| =", abs(a[0][-1]))
