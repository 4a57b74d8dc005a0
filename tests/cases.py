"""Shared helpers: load a golden fixture and build the oracle objects for it."""
import os

import numpy as np
import torch

from oracle import sc_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

HK_CASES = ["hk_1d", "hk_as5_chi000", "hk_as5_chi002", "hk_as60", "hk_as60_dt20", "hk_methylium",
            "hk_as60_n96", "hk_as33"]          # round 4 (tests/golden/make_golden_round4.py): more trajectories / steps at D = 60, D = 33
GDML_CASES = ["hk_coumarin_gdml"]
WM_CASES = ["wm_1d", "wm_as5_chi002", "wm_methylium", "wm_as24", "wm_as60"]          # wm_as24, wm_as60: D > 16 (round 4)


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def T(x):
    return torch.from_numpy(np.asarray(x)).clone()


def oracle_potential(g):
    kind = str(g["potential"])
    if kind == "morse":
        return orc.MorseOracle(g["omega"], g["chi"], g["nac"])
    if kind == "nonharmonic":
        return orc.NonHarmonicOracle(g["eps"], g["b"])
    if kind == "harmonic":
        return orc.MolecularHarmonicOracle(g["pos0"], g["energy0"], g["grad0"], g["hess0"], g["masses"],
                                           g["nac0"], origin=float(g["origin"]))
    if kind == "gdml":
        return orc.MolecularGDMLOracle(load("gdml_coumarin_model"), g["masses"], g["nac0"], origin=float(g["origin"]))
    raise ValueError(kind)


def oracle_propagator(g):
    Gi, Gt = T(g["Gamma_i"]), T(g["Gamma_t"])
    if "alpha" in g:
        prop = orc.WMOracle(Gi, Gt, float(g["alpha"]), float(g["beta"]))
    else:
        prop = orc.HKOracle(Gi, Gt)
    prop.set_initial_conditions(T(g["q0"]), T(g["p0"]), T(g["Gamma_0"]), T(g["zi"]), T(g["probi"]))
    return prop


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
