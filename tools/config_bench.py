#!/usr/bin/env python
"""Throughput of the other BASELINE.json configurations (not the bench line): GPU box only.

    python tools/config_bench.py [ntraj] [steps]

config 1: 5-mode anharmonic AS, HK      config 3: methylium (D=12, rank-6 Gamma_0), harmonic Cartesian potential, WM
Inputs come from the committed golden fixtures (parameters only); initial conditions are sampled afresh.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import cases  # noqa: E402
from tests.engine_cases import engine_potential  # noqa: E402
from semiclassical_amd import propagators as PR  # noqa: E402

torch.set_default_dtype(torch.float64)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50


def run(name, label):
    g = cases.load(name)
    pot = engine_potential(g)
    Gi, Gt = cases.T(g["Gamma_i"]), cases.T(g["Gamma_t"])
    if "alpha" in g:
        prop = PR.WaltonManolopoulosPropagator(Gi, Gt, float(g["alpha"]), float(g["beta"]), device="cuda")
    else:
        prop = PR.HermanKlukPropagator(Gi, Gt, device="cuda")
    prop.initial_conditions(cases.T(g["q0"]), cases.T(g["p0"]), cases.T(g["Gamma_0"]), ntraj=n,
                            generator=torch.Generator().manual_seed(7))
    dt, E0 = float(g["dt"]), float(g["E0"])
    prop.run(pot, dt, 3, E0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c, k = prop.run(pot, dt, steps, E0)
    wall = time.perf_counter() - t0
    assert np.isfinite(c).all() and np.isfinite(k).all()
    print(f"{label}: D={prop.dim} n={n} steps={steps}  {n * steps / wall:.3e} trajectory-steps/s  "
          f"({wall / steps * 1e3:.3f} ms/step)  |C(0)|={abs(c[0]):.4f}", flush=True)


def run_gdml30(n, steps):
    """config 5 at its stated size: a synthetic 30-atom sGDML model (no such model ships with the reference), D = 90"""
    from tests.test_gdml_gpu import synthetic_model
    from oracle import sc_oracle as orc
    from semiclassical_amd.gdml import MolecularGDMLPotential
    N = 30
    model, pos = synthetic_model(N, 200, 30)
    g0 = orc.GDMLOracle(model).forward(torch.from_numpy(pos.reshape(1, -1)))[1]
    model["R_d_desc_alpha"] = model["R_d_desc_alpha"] * (0.02 / float(g0.abs().max()))
    masses = np.repeat(np.full(N, 12.0 * 1822.888), 3)

    class _Fchk(object):
        def nonadiabatic_coupling(self_):
            return np.zeros(3 * N)

        def masses(self_):
            return masses

        def atomic_numbers(self_):
            return model["z"]
    pot = MolecularGDMLPotential(model, _Fchk())
    q0 = torch.from_numpy(pos.reshape(-1))
    G = torch.diag(torch.full((3 * N,), 40.0))
    prop = PR.HermanKlukPropagator(G, G, device="cuda")
    prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, generator=torch.Generator().manual_seed(7))
    prop.run(pot, 2.0, 2, 0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c, k = prop.run(pot, 2.0, steps, 0.0)
    wall = time.perf_counter() - t0
    print(f"config 5  HK  synthetic 30-atom sGDML (M=200): D=90 n={n} steps={steps}  {n * steps / wall:.3e} "
          f"trajectory-steps/s  ({wall / steps * 1e3:.3f} ms/step)", flush=True)


if os.environ.get("GDML30"):
    run_gdml30(int(sys.argv[1]) if len(sys.argv) > 1 else 10000, int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    sys.exit(0)
if os.environ.get("GDML_ONLY"):
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    run("hk_coumarin_gdml", "config 5* HK  coumarin sGDML (17 atoms, M=200)")
    sys.exit(0)
run("hk_as5_chi002", "config 1  HK  5-mode anharmonic AS")
run("wm_as5_chi002", "          WM  5-mode anharmonic AS")
run("hk_methylium", "          HK  methylium harmonic")
run("wm_methylium", "config 3  WM  methylium harmonic")
