#!/usr/bin/env python
"""Where does the host time of one run() iteration go?  (GPU box; cProfile of the shortcut path, which is launch-bound)"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from semiclassical_amd import potentials as P, propagators as PR  # noqa: E402

torch.set_default_dtype(torch.float64)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
omega, chi, nac, q0, dt = bench.as60_model()
G = torch.diag(omega)
pot = P.MorsePotential(omega, chi.clone(), nac)
prop = PR.HermanKlukPropagator(G, G, device="cuda", exploit_separability=True)
prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, generator=torch.Generator().manual_seed(1))
slots = torch.zeros((200, 5), device="cuda")
prop.run(pot, dt, 5, 0.0, slots=slots)
prop.synchronize()
t0 = time.perf_counter()
prop.run(pot, dt, 200, 0.0, slots=slots)
t1 = time.perf_counter()
prop.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / 200:.3f} ms/step, total {1e3 * (t2 - t0) / 200:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
prop.run(pot, dt, 200, 0.0, slots=slots)
pr.disable()
prop.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
