"""Build semiclassical_amd (HIP engine) objects from a golden fixture."""
import torch

from tests import cases


def engine_potential(g):
    from semiclassical_amd import potentials as P
    kind = str(g["potential"])
    if kind == "morse":
        return P.MorsePotential(cases.T(g["omega"]), cases.T(g["chi"]), cases.T(g["nac"]))
    if kind == "nonharmonic":
        return P.NonHarmonicPotential(cases.T(g["eps"]), cases.T(g["b"]))
    if kind == "harmonic":
        return P.MolecularHarmonicPotential.from_arrays(g["pos0"], g["energy0"], g["grad0"], g["hess0"],
                                                        g["masses"], g["nac0"], origin=float(g["origin"]))
    if kind == "gdml":
        from semiclassical_amd.gdml import MolecularGDMLPotential

        class _Fchk(object):
            def nonadiabatic_coupling(self_):
                return g["nac0"]

            def masses(self_):
                return g["masses"]

            def atomic_numbers(self_):
                return model["z"]
        model = cases.load("gdml_coumarin_model")
        pot = MolecularGDMLPotential(model, _Fchk())
        pot._origin = float(g["origin"])
        pot._invalidate_descriptor()
        return pot
    raise ValueError(kind)


def engine_propagator(g, device="cuda", select=None, ntraj_total=None, **kwargs):
    """``select``: a slice of the fixture's trajectories (one rank's shard), ``ntraj_total``: the N of their weights"""
    from semiclassical_amd import propagators as PR
    Gi, Gt = cases.T(g["Gamma_i"]), cases.T(g["Gamma_t"])
    if "alpha" in g:
        prop = PR.WaltonManolopoulosPropagator(Gi, Gt, float(g["alpha"]), float(g["beta"]), device=device)
    else:
        prop = PR.HermanKlukPropagator(Gi, Gt, device=device, **kwargs)
    sel = slice(None) if select is None else select
    prop.set_initial_conditions(cases.T(g["q0"]), cases.T(g["p0"]), cases.T(g["Gamma_0"]),
                                cases.T(g["zi"][:, sel]), cases.T(g["probi"][sel]), ntraj_total=ntraj_total)
    return prop
