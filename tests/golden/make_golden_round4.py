#!/usr/bin/env python
"""Round-4 additions to the golden vectors, produced by running the REFERENCE itself (build container only; see make_golden.py for the
compatibility aliases and for what a file holds):

  hk_as60_n96   the synthetic 60-mode anharmonic AS model of BASELINE.json config 2 with more trajectories and more, larger steps than
                hk_as60 / hk_as60_dt20 (96 trajectories, 40 steps of 10 x the benchmark's dt): the case the two-steps-per-launch path of
                run() is checked against
  hk_as33       the first 33 modes of the same model (three 16-row slots in the fast kernel), 64 trajectories, 30 steps
  wm_as24       Walton-Manolopoulos on the first 24 modes (D > 16: the LDS / global-scratch WM kernel), 24 trajectories, 8 steps
  wm_as60       Walton-Manolopoulos on all 60 modes, 12 trajectories, 4 steps
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402  (imports the reference, defines run_case / as_model / synthetic_as60)
from semiclassical.propagators import HermanKlukPropagator, WaltonManolopoulosPropagator  # noqa: E402
from semiclassical.potentials import MorsePotential                                      # noqa: E402
from semiclassical import units                                                          # noqa: E402


def model(modes):
    omega_cm, S, nac, chi = mg.synthetic_as60()
    omega, dQ, nac, chi = mg.as_model(omega_cm[:modes], S[:modes], nac[:modes], chi[:modes])
    ex = dict(potential="morse", omega=omega.numpy(), chi=chi.numpy().copy(), nac=nac.numpy())
    return omega, dQ, MorsePotential(omega, chi.clone(), nac), torch.diag(omega), torch.sum(0.5 * omega).item(), ex


def main():
    dt = 0.005 / units.autime_to_fs
    omega, dQ, pot, G, E0, ex = model(60)
    mg.run_case("hk_as60_n96", lambda: HermanKlukPropagator(G, G), pot, dQ, 0.0 * dQ, G, 10 * dt, 40, E0, 96, [1, 20, 40], ex, store_y=False)
    omega, dQ, pot, G, E0, ex = model(33)
    mg.run_case("hk_as33", lambda: HermanKlukPropagator(G, G), pot, dQ, 0.0 * dQ, G, 20 * dt, 30, E0, 64, [1, 30], ex, store_y=False)
    omega, dQ, pot, G, E0, ex = model(24)
    mg.run_case("wm_as24", lambda: WaltonManolopoulosPropagator(G, G, 500, 500), pot, dQ, 0.0 * dQ, G, 20 * dt, 8, E0, 24, [1, 8],
                dict(ex, alpha=500.0, beta=500.0), store_y=False)
    omega, dQ, pot, G, E0, ex = model(60)
    mg.run_case("wm_as60", lambda: WaltonManolopoulosPropagator(G, G, 500, 500), pot, dQ, 0.0 * dQ, G, 20 * dt, 4, E0, 12, [1, 4],
                dict(ex, alpha=500.0, beta=500.0), store_y=False)


if __name__ == "__main__":
    main()
