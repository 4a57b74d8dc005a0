#!/usr/bin/env python
"""Golden values of HermanKlukPropagator.norm() / coefficients() from the REFERENCE.

Build container only.  Uses the initial conditions (zi, probi) of existing golden cases, so the engine and the
oracle can be started from identical states; stores the norm and the expansion coefficients after `nsteps` steps.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (torch aliases + reference import)
from semiclassical.propagators import HermanKlukPropagator, WaltonManolopoulosPropagator  # noqa: E402
from semiclassical.potentials import MorsePotential, MolecularHarmonicPotential  # noqa: E402


def run(name, potential, nsteps):
    g = dict(np.load(os.path.join(HERE, name + ".npz")))
    T = lambda x: torch.from_numpy(np.asarray(x)).clone()
    torch.manual_seed(0)
    prop = HermanKlukPropagator(T(g["Gamma_i"]), T(g["Gamma_t"]))
    prop.initial_conditions(T(g["q0"]), T(g["p0"]), T(g["Gamma_0"]), ntraj=g["zi"].shape[1])
    assert np.array_equal(prop.zi.numpy(), g["zi"])          # same seed => same initial conditions as the golden
    # spatial grid for wavefunction(): points on the segment between the centre q0 and the mean final position, and beyond
    rng = np.random.default_rng(7)
    d = g["q0"].shape[0]
    xgrid = g["q0"][:, None] + 0.3 * rng.standard_normal((d, 9)) / np.sqrt(np.maximum(np.diag(g["Gamma_t"]), 1e-3))[:, None]
    out = {"norm_0": prop.norm(), "coeff_0": prop.coefficients().numpy(), "xgrid": xgrid,
           "psi_0": prop.wavefunction(T(xgrid))}
    for _ in range(nsteps):
        prop.step(potential, float(g["dt"]))
    out.update({"nsteps": nsteps, f"norm_{nsteps}": prop.norm(), f"coeff_{nsteps}": prop.coefficients().numpy(),
                f"psi_{nsteps}": prop.wavefunction(T(xgrid))})
    print(name, out["norm_0"], out[f"norm_{nsteps}"])
    return out


def run_wm(name, potential, nsteps, with_norm):
    """WM: coefficients, wavefunction (and, for small cases, the O(n^2 d'^3) norm) at step 0 and after nsteps"""
    g = dict(np.load(os.path.join(HERE, name + ".npz")))
    T = lambda x: torch.from_numpy(np.asarray(x)).clone()
    torch.manual_seed(0)
    prop = WaltonManolopoulosPropagator(T(g["Gamma_i"]), T(g["Gamma_t"]), float(g["alpha"]), float(g["beta"]))
    prop.initial_conditions(T(g["q0"]), T(g["p0"]), T(g["Gamma_0"]), ntraj=g["zi"].shape[1])
    assert np.array_equal(prop.zi.numpy(), g["zi"])
    rng = np.random.default_rng(11)
    d = g["q0"].shape[0]
    xgrid = g["q0"][:, None] + 0.3 * rng.standard_normal((d, 9)) / np.sqrt(np.maximum(np.diag(g["Gamma_t"]), 1e-3))[:, None]
    out = {"xgrid": xgrid, "coeff_0": prop.coefficients().numpy(), "psi_0": prop.wavefunction(T(xgrid))}
    if with_norm:
        out["norm_0"] = prop.norm()
    for _ in range(nsteps):
        prop.step(potential, float(g["dt"]))
    out.update({"nsteps": nsteps, f"coeff_{nsteps}": prop.coefficients().numpy(), f"psi_{nsteps}": prop.wavefunction(T(xgrid))})
    if with_norm:
        out[f"norm_{nsteps}"] = prop.norm()
    print(name, abs(out["psi_0"]).max(), out.get("norm_0"), out.get(f"norm_{nsteps}"))
    return out


def main_wm():
    from semiclassical.potentials import NonHarmonicPotential
    g = dict(np.load(os.path.join(HERE, "wm_1d.npz")))
    res = {"wm1d_" + k: v for k, v in run_wm("wm_1d", NonHarmonicPotential(), 20, True).items()}
    g = dict(np.load(os.path.join(HERE, "wm_as5_chi002.npz")))
    pot = MorsePotential(torch.from_numpy(g["omega"]), torch.from_numpy(g["chi"]).clone(), torch.from_numpy(g["nac"]))
    res.update({"wmas5_" + k: v for k, v in run_wm("wm_as5_chi002", pot, 10, True).items()})
    g = dict(np.load(os.path.join(HERE, "wm_methylium.npz")))
    fchk = mg._Fchk(pos0=g["pos0"], energy0=g["energy0"], grad0=g["grad0"], hess0=g["hess0"], _m=g["masses"], nac0=g["nac0"])
    pot = MolecularHarmonicPotential(fchk, fchk)
    pot._origin = float(g["origin"])
    res.update({"wmmet_" + k: v for k, v in run_wm("wm_methylium", pot, 5, True).items()})
    np.savez_compressed(os.path.join(HERE, "wm_norms.npz"), **res)


def main():
    g = dict(np.load(os.path.join(HERE, "hk_as5_chi002.npz")))
    pot = MorsePotential(torch.from_numpy(g["omega"]), torch.from_numpy(g["chi"]).clone(), torch.from_numpy(g["nac"]))
    res = {"as5_" + k: v for k, v in run("hk_as5_chi002", pot, 20).items()}
    g = dict(np.load(os.path.join(HERE, "hk_methylium.npz")))
    fchk = mg._Fchk(pos0=g["pos0"], energy0=g["energy0"], grad0=g["grad0"], hess0=g["hess0"], _m=g["masses"], nac0=g["nac0"])
    pot = MolecularHarmonicPotential(fchk, fchk)
    pot._origin = float(g["origin"])
    res.update({"met_" + k: v for k, v in run("hk_methylium", pot, 10).items()})
    np.savez_compressed(os.path.join(HERE, "hk_norms.npz"), **res)


if __name__ == "__main__":
    if "wm" in sys.argv[1:]:
        main_wm()
    else:
        main()
