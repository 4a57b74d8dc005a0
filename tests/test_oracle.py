"""The CPU oracle against the golden vectors produced by the reference itself."""
import numpy as np
import pytest
import torch

from oracle import sc_oracle as orc
from tests import cases

torch.set_default_dtype(torch.float64)

TOL = 1e-11


@pytest.mark.parametrize("name", cases.HK_CASES + cases.WM_CASES + cases.GDML_CASES)
def test_oracle_matches_reference(name):
    g = cases.load(name)
    pot = cases.oracle_potential(g)
    prop = cases.oracle_propagator(g)
    nt, dt, E0 = int(g["nt"]), float(g["dt"]), float(g["E0"])
    is_wm = "alpha" in g
    assert cases.rel_err(prop.U.numpy(), g["U"]) < 1e-14
    assert cases.rel_err(prop.iGi0.numpy(), g["iGi0"]) < 1e-13
    cauto = np.zeros(nt, dtype=complex)
    kic = np.zeros(nt, dtype=complex)
    for t in range(nt):
        assert cases.rel_err(prop.c2.numpy(), g["c2"][t]) < TOL
        cauto[t] = prop.autocorrelation(E0)
        kic[t] = prop.ic_correlation(pot, E0)
        prop.step(pot, dt)
        step = t + 1
        if step in g["snaps"]:
            if f"y_{step}" in g:
                assert cases.rel_err(prop.y.numpy(), g[f"y_{step}"]) < TOL
            else:
                assert cases.rel_err(prop.y[:, 0].numpy(), g[f"ytraj0_{step}"]) < TOL
            assert np.array_equal(prop.tracker.signs("prefactorC").numpy(), g[f"signs_{step}"])
            assert cases.rel_err(prop.autocorrelation_qp().numpy(), g[f"cauto_qp_{step}"]) < 1e-10
            if is_wm:
                assert np.array_equal(prop.tracker.signs("detA").numpy(), g[f"signsA_{step}"])
                assert np.array_equal(prop.tracker.signs("detM").numpy(), g[f"signsM_{step}"])
                assert cases.rel_err(prop.gamma.numpy(), g[f"gamma_{step}"]) < 1e-10
    assert cases.rel_err(cauto, g["cauto"]) < TOL
    assert cases.rel_err(kic, g["kic"]) < 1e-10


def test_seeded_sampling_reproduces_reference():
    """initial_conditions() under manual_seed(0) draws the reference's zi/probi (same torch CPU RNG)."""
    g = cases.load("hk_as5_chi002")
    prop = orc.HKOracle(cases.T(g["Gamma_i"]), cases.T(g["Gamma_t"]))
    torch.manual_seed(0)
    prop.initial_conditions(cases.T(g["q0"]), cases.T(g["p0"]), cases.T(g["Gamma_0"]), ntraj=g["zi"].shape[1])
    if abs(prop.zi[0, 0].item() - g["zi"][0, 0]) > 1e-9:
        pytest.skip("torch CPU normal stream differs on this host")
    assert cases.rel_err(prop.zi.numpy(), g["zi"]) < 1e-14
    assert cases.rel_err(prop.probi.numpy(), g["probi"]) < 1e-13


def test_overlap_normalisation_and_zero_modes():
    """reference tests/test_propagators.py:73-113"""
    torch.manual_seed(0)
    n = 5
    Gi = 5.0 * 2.0 * (torch.rand(n, n) - 0.5)
    Gi = 0.5 * (Gi + Gi.T)
    qi, pi = torch.rand(n, 1), torch.rand(n, 1)
    olap = orc.OverlapOracle(Gi, Gi)(qi, pi, qi, pi).squeeze().item()
    assert abs(olap - 1.0) < 1e-5
    Gi_ = torch.zeros((n + 1, n + 1))
    Gi_[:n, :n] = Gi
    qi_, pi_ = (torch.cat((x, torch.zeros(1, 1)), 0) for x in (qi, pi))
    olap_ = orc.OverlapOracle(Gi_, Gi_)(qi_, pi_, qi_, pi_).squeeze().item()
    assert olap == olap_


def test_sym_sqrtm():
    """reference tests/test_propagators.py:40-54"""
    import scipy.linalg as sla
    torch.manual_seed(0)
    A = 5.0 * 2.0 * (torch.rand(5, 5) - 0.5)
    A = A + A.T
    s, i = orc.sym_sqrtm(A)
    assert np.allclose(s.numpy(), sla.sqrtm(A.numpy()))
    assert np.allclose(i.numpy(), sla.inv(sla.sqrtm(A.numpy())))


def test_gdml_oracle_matches_reference_predictor():
    """E, grad, Hessian of the sGDML restatement vs GDMLPredict.forward of the reference (golden)"""
    g = cases.load("gdml_coumarin_eval")
    gd = orc.GDMLOracle(cases.load("gdml_coumarin_model"))
    e, grad, hess = gd.forward(torch.from_numpy(g["r"]))
    assert cases.rel_err(e.numpy(), g["energy"]) < 1e-14
    assert cases.rel_err(grad.numpy(), g["grad"]) < 1e-12
    assert cases.rel_err(hess.numpy(), g["hess"]) < 1e-12
    assert torch.allclose(hess, hess.transpose(1, 2), atol=1e-10)        # reference tests/test_gdml_predictor.py:90-122


def test_gdml_sum_conditioning():
    """why GPU-vs-reference agreement on sGDML forces is ~1e-8, not 1e-13: the reference's own formula changes by
    that much when the training points are summed in a different order (terms of 2e8 cancel to 6e1)"""
    g = cases.load("gdml_coumarin_eval")
    gd = orc.GDMLOracle(cases.load("gdml_coumarin_model"))
    r = torch.from_numpy(g["r"])
    _, grad, hess = gd.forward(r)
    perm = torch.randperm(gd.xs_train.shape[0], generator=torch.Generator().manual_seed(0))
    gd.xs_train, gd.Jx_alphas = gd.xs_train[perm], gd.Jx_alphas[perm]
    _, grad2, hess2 = gd.forward(r)
    assert 1e-10 < cases.rel_err(grad2.numpy(), grad.numpy()) < 1e-6
    assert cases.rel_err(hess2.numpy(), hess.numpy()) < 1e-6


@pytest.mark.parametrize("name,tag", [("hk_as5_chi002", "as5"), ("hk_methylium", "met")])
def test_norm_oracle_matches_reference(name, tag):
    from oracle import norm_oracle
    g, ref = cases.load(name), cases.load("hk_norms")
    pot, prop = cases.oracle_potential(g), cases.oracle_propagator(g)
    assert abs(norm_oracle.norm(prop) - float(ref[f"{tag}_norm_0"])) < 1e-12
    assert cases.rel_err(norm_oracle.wavefunction(prop, ref[f"{tag}_xgrid"]), ref[f"{tag}_psi_0"]) < 1e-11
    for _ in range(int(ref[f"{tag}_nsteps"])):
        prop.step(pot, float(g["dt"]))
    n = int(ref[f"{tag}_nsteps"])
    assert cases.rel_err(norm_oracle.coefficients(prop).numpy(), ref[f"{tag}_coeff_{n}"]) < 1e-11
    assert abs(norm_oracle.norm(prop) - float(ref[f"{tag}_norm_{n}"])) < 1e-10
    assert cases.rel_err(norm_oracle.wavefunction(prop, ref[f"{tag}_xgrid"]), ref[f"{tag}_psi_{n}"]) < 1e-10


@pytest.mark.parametrize("name,tag", [("wm_1d", "wm1d"), ("wm_as5_chi002", "wmas5"), ("wm_methylium", "wmmet")])
def test_wm_coefficients_and_wavefunction_oracle_matches_reference(name, tag):
    from oracle import norm_oracle
    g, ref = cases.load(name), cases.load("wm_norms")
    pot, prop = cases.oracle_potential(g), cases.oracle_propagator(g)
    x = ref[f"{tag}_xgrid"]
    assert cases.rel_err(norm_oracle.wm_coefficients(prop).numpy(), ref[f"{tag}_coeff_0"]) < 1e-11
    assert cases.rel_err(norm_oracle.wm_wavefunction(prop, x), ref[f"{tag}_psi_0"]) < 1e-11
    n = int(ref[f"{tag}_nsteps"])
    for _ in range(n):
        prop.step(pot, float(g["dt"]))
    assert cases.rel_err(norm_oracle.wm_coefficients(prop).numpy(), ref[f"{tag}_coeff_{n}"]) < 1e-9
    assert cases.rel_err(norm_oracle.wm_wavefunction(prop, x), ref[f"{tag}_psi_{n}"]) < 1e-9
    want = float(ref[f"{tag}_norm_{n}"])
    assert abs(norm_oracle.wm_norm(prop) - want) < 1e-9 * want
