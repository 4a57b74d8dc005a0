#!/usr/bin/env python
"""Time of one sGDML stage launch (sc_gdml_stage) on the coumarin model (GPU box; A/B tool)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases  # noqa: E402
from tests.engine_cases import engine_potential  # noqa: E402
from semiclassical_amd import propagators as PR  # noqa: E402
from semiclassical_amd._lib import lib, check, ptr  # noqa: E402

torch.set_default_dtype(torch.float64)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
if os.environ.get("N30"):               # the synthetic 30-atom model of config 5 (D = 90)
    import numpy as np
    from semiclassical_amd.gdml import MolecularGDMLPotential
    from semiclassical_amd.synthetic import sgdml_model, ArrayFchk
    model_, pos = sgdml_model(30, 200, 30)
    pot = MolecularGDMLPotential(model_, ArrayFchk(np.repeat(np.full(30, 12.0 * 1822.888), 3), np.zeros(90), model_["z"]))
    q0 = torch.from_numpy(pos.reshape(-1))
    G = torch.diag(torch.full((90,), 40.0))
    prop = PR.HermanKlukPropagator(G, G, device="cuda")
    prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, generator=torch.Generator().manual_seed(7))
    prop.step(pot, 0.1)
    name = "30 atoms"
else:
    g = cases.load("hk_coumarin_gdml")
    pot = engine_potential(g)
    prop = PR.HermanKlukPropagator(cases.T(g["Gamma_i"]), cases.T(g["Gamma_t"]), device="cuda")
    prop.initial_conditions(cases.T(g["q0"]), cases.T(g["p0"]), cases.T(g["Gamma_0"]), ntraj=n, generator=torch.Generator().manual_seed(7))
    prop.step(pot, float(g["dt"]))          # allocates the dense scratch
    name = "coumarin"
model = pot._gdml_model(prop.device)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    check(lib.sc_gdml_stage(model, prop._state, prop._dense, 0.0, 0, ptr(prop._epart), prop._stream()))
e1.record()
torch.cuda.synchronize()
print(f"{name} n={n}: sGDML stage kernel {e0.elapsed_time(e1) / 5:.3f} ms", flush=True)
