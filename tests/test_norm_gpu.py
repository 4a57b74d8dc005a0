"""Row N1: coefficients(), norm() and wavefunction() of the HK propagator against values produced by the reference."""
import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu
torch.set_default_dtype(torch.float64)      # the oracle follows the reference's global default (cli.py:121)


@pytest.mark.parametrize("name,tag", [("hk_as5_chi002", "as5"), ("hk_methylium", "met")])
def test_norm_and_coefficients_match_reference(name, tag):
    from tests.engine_cases import engine_potential, engine_propagator
    g, ref = cases.load(name), cases.load("hk_norms")
    pot, prop = engine_potential(g), engine_propagator(g)
    nsteps = int(ref[f"{tag}_nsteps"])
    assert abs(prop.norm() - float(ref[f"{tag}_norm_0"])) < 1e-9 * float(ref[f"{tag}_norm_0"])
    assert cases.rel_err(prop.coefficients().cpu().numpy(), ref[f"{tag}_coeff_0"]) < 1e-10
    assert cases.rel_err(prop.wavefunction(cases.T(ref[f"{tag}_xgrid"])), ref[f"{tag}_psi_0"]) < 1e-10
    for _ in range(nsteps):
        prop.step(pot, float(g["dt"]))
    assert cases.rel_err(prop.coefficients().cpu().numpy(), ref[f"{tag}_coeff_{nsteps}"]) < 1e-9
    want = float(ref[f"{tag}_norm_{nsteps}"])
    assert abs(prop.norm() - want) < 1e-9 * want
    assert cases.rel_err(prop.wavefunction(cases.T(ref[f"{tag}_xgrid"])), ref[f"{tag}_psi_{nsteps}"]) < 1e-9


def test_wavefunction_matches_oracle_on_a_ragged_grid():
    """nx not a multiple of the grid tile, many trajectories: engine vs the CPU oracle on the same state"""
    from oracle import norm_oracle
    from tests.engine_cases import engine_potential, engine_propagator
    g = cases.load("hk_as5_chi002")
    pot, prop = engine_potential(g), engine_propagator(g)
    opot, oprop = cases.oracle_potential(g), cases.oracle_propagator(g)
    for _ in range(5):
        prop.step(pot, float(g["dt"]))
        oprop.step(opot, float(g["dt"]))
    x = cases.T(g["q0"])[:, None] + 0.5 * torch.randn((5, 13), generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    assert cases.rel_err(prop.wavefunction(x), norm_oracle.wavefunction(oprop, x)) < 1e-9
    with pytest.raises(AssertionError):
        prop.wavefunction(x[:4])


def test_norm_of_many_trajectories_is_one():
    """reference tests/test_propagators.py:299: |psi| ~ 1 once the basis of coherent states is large enough"""
    from tests.engine_cases import engine_potential
    from semiclassical_amd import propagators as PR
    g = cases.load("hk_1d")
    pot = engine_potential(g)
    Gi = cases.T(g["Gamma_i"])
    prop = PR.HermanKlukPropagator(Gi, Gi, device="cuda")
    prop.initial_conditions(cases.T(g["q0"]), cases.T(g["p0"]), cases.T(g["Gamma_0"]), ntraj=50000,
                            generator=torch.Generator().manual_seed(0))
    for _ in range(20):
        prop.step(pot, float(g["dt"]))
    assert abs(prop.norm() - 1.0) < 0.05
