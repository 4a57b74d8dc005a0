"""Parity of the HIP Walton-Manolopoulos path with the reference's golden vectors (through the C-ABI)."""
import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu
TOL = 1e-8     # asserted; north_star requires 1e-6


def cnp(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("name", cases.WM_CASES)
def test_wm_matches_reference_golden(name):
    from tests.engine_cases import engine_potential, engine_propagator
    g = cases.load(name)
    pot = engine_potential(g)
    prop = engine_propagator(g)
    nt, dt, E0 = int(g["nt"]), float(g["dt"]), float(g["E0"])
    cauto = np.zeros(nt, dtype=complex)
    kic = np.zeros(nt, dtype=complex)
    for t in range(nt):
        assert cases.rel_err(cnp(prop._detA), g["detA"][t]) < TOL, f"detA at step {t}"
        assert cases.rel_err(cnp(prop._detM), g["detM"][t]) < TOL, f"detM at step {t}"
        cauto[t] = prop.autocorrelation(E0)
        kic[t] = prop.ic_correlation(pot, E0)
        prop.step(pot, dt)
        step = t + 1
        if step in g["snaps"]:
            assert np.array_equal(cnp(prop._sgnA), g[f"signsA_{step}"].real)
            assert np.array_equal(cnp(prop._sgnM), g[f"signsM_{step}"].real)
            assert cases.rel_err(cnp(prop.autocorrelation_qp()), g[f"cauto_qp_{step}"]) < TOL
    prop.synchronize()
    assert cases.rel_err(cauto, g["cauto"]) < TOL
    assert cases.rel_err(kic, g["kic"]) < TOL


@pytest.mark.parametrize("name", ["wm_methylium", "wm_as5_chi002"])
def test_wm_fused_run(name):
    from tests.engine_cases import engine_potential, engine_propagator
    g = cases.load(name)
    prop = engine_propagator(g)
    cauto, kic = prop.run(engine_potential(g), float(g["dt"]), int(g["nt"]), float(g["E0"]))
    assert cases.rel_err(cauto, g["cauto"]) < TOL
    assert cases.rel_err(kic, g["kic"]) < TOL


@pytest.mark.parametrize("name,tag", [("wm_1d", "wm1d"), ("wm_as5_chi002", "wmas5"), ("wm_methylium", "wmmet")])
def test_wm_coefficients_and_wavefunction_match_reference(name, tag):
    """rest of row N1 for WM: coefficients() (eqn 75) and wavefunction() against values produced by the reference"""
    from tests.engine_cases import engine_potential, engine_propagator
    g, ref = cases.load(name), cases.load("wm_norms")
    pot, prop = engine_potential(g), engine_propagator(g)
    x = cases.T(ref[f"{tag}_xgrid"])
    assert cases.rel_err(prop.coefficients().cpu().numpy(), ref[f"{tag}_coeff_0"]) < 1e-9
    assert cases.rel_err(prop.wavefunction(x), ref[f"{tag}_psi_0"]) < 1e-9
    assert abs(prop.norm() - float(ref[f"{tag}_norm_0"])) < 1e-8 * float(ref[f"{tag}_norm_0"])
    n = int(ref[f"{tag}_nsteps"])
    for _ in range(n):
        prop.step(pot, float(g["dt"]))
    assert cases.rel_err(prop.coefficients().cpu().numpy(), ref[f"{tag}_coeff_{n}"]) < 1e-8
    assert cases.rel_err(prop.wavefunction(x), ref[f"{tag}_psi_{n}"]) < 1e-8
    assert abs(prop.norm() - float(ref[f"{tag}_norm_{n}"])) < 1e-8 * float(ref[f"{tag}_norm_{n}"])
    # the export launch must not disturb the correlation functions of the same step
    c1 = prop.autocorrelation(float(g["E0"]))
    prop._wm_export_step = -1
    prop.coefficients()
    prop._corr_step = -1
    assert prop.autocorrelation(float(g["E0"])) == c1


def test_wm_norm_of_many_trajectories_is_one():
    """reference tests/test_propagators.py:302-327: |psi| ~ 1 for the 1-D WM wavepacket with enough trajectories"""
    from tests.engine_cases import engine_potential
    from semiclassical_amd import propagators as PR
    g = cases.load("wm_1d")
    pot = engine_potential(g)
    Gi = cases.T(g["Gamma_i"])
    prop = PR.WaltonManolopoulosPropagator(Gi, Gi, float(g["alpha"]), float(g["beta"]), device="cuda")
    prop.initial_conditions(cases.T(g["q0"]), cases.T(g["p0"]), cases.T(g["Gamma_0"]), ntraj=20000,
                            generator=torch.Generator().manual_seed(0))
    for _ in range(20):
        prop.step(pot, float(g["dt"]))
    assert abs(prop.norm() - 1.0) < 0.05
