"""sGDML row on the GPU: function-level parity with GDMLPredict.forward and a short
HK propagation on the coumarin surface, both against vectors produced by the reference."""
import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu


def cnp(t):
    return t.detach().cpu().numpy()


def test_gdml_energy_gradient_hessian_match_reference():
    from tests.engine_cases import engine_potential
    g = cases.load("gdml_coumarin_eval")
    pot = engine_potential(dict(potential="gdml", nac0=np.zeros(51), masses=np.ones(51), origin=0.0))
    r = torch.from_numpy(g["r"]).t().contiguous().cuda()                 # (D, n)
    v, grad, hess = pot.harmonic_approximation(r)
    assert cases.rel_err(cnp(v), g["energy"]) < 1e-11      # E = std*sum + c cancels 1e4 Hartree
    # the sGDML sums cancel terms of 2e8 down to 6e1: re-ordering the training points in the reference's own
    # formula already moves grad by 2e-8 and hess by 2e-9 (tests/test_oracle.py::test_gdml_sum_conditioning)
    assert cases.rel_err(cnp(grad.t()), g["grad"]) < 2e-7
    assert cases.rel_err(cnp(hess.permute(2, 0, 1)), g["hess"]) < 2e-7
    h = cnp(hess.permute(2, 0, 1))
    assert np.max(np.abs(h - h.transpose(0, 2, 1))) < 1e-10 * np.max(np.abs(h))   # reference tests/test_gdml_predictor.py:90-122


def test_hk_on_gdml_surface_matches_reference_golden():
    from tests.engine_cases import engine_potential, engine_propagator
    g = cases.load("hk_coumarin_gdml")
    pot = engine_potential(g)
    prop = engine_propagator(g)
    nt, dt, E0 = int(g["nt"]), float(g["dt"]), float(g["E0"])
    cauto = np.zeros(nt, dtype=complex)
    kic = np.zeros(nt, dtype=complex)
    for t in range(nt):
        assert cases.rel_err(cnp(prop._c2), g["c2"][t]) < 1e-6, f"c2 at step {t}"
        cauto[t] = prop.autocorrelation(E0)
        kic[t] = prop.ic_correlation(pot, E0)
        prop.step(pot, dt)
        step = t + 1
        if step in g["snaps"]:
            y = cnp(prop.y)
            d = prop.dim
            assert cases.rel_err(np.vstack((y[:2 * d], y[-1:])), g[f"qpS_{step}"]) < 1e-7
            assert cases.rel_err(y[:, 0], g[f"ytraj0_{step}"]) < 1e-7
            assert np.array_equal(cnp(prop._sgn), g[f"signs_{step}"].real)
    prop.synchronize()
    assert cases.rel_err(cauto, g["cauto"]) < 1e-6        # north_star tolerance (force conditioning, see above)
    assert cases.rel_err(kic, g["kic"]) < 1e-6


def synthetic_model(n_atoms, n_train, seed):
    """an sGDML model of the right shapes for a molecule the reference ships no model for (SURVEY.md section 8d,
    config 5: 30 atoms): atoms on a jittered lattice, training descriptors = descriptors of perturbed geometries,
    coefficients scaled like the coumarin model's"""
    rng = np.random.default_rng(seed)
    side = int(np.ceil(n_atoms ** (1 / 3)))
    grid = np.array([(i, j, k) for i in range(side) for j in range(side) for k in range(side)], dtype=float)[:n_atoms]
    pos = 2.6 * grid + rng.normal(0, 0.15, grid.shape)
    k, l = np.tril_indices(n_atoms, -1)
    desc = lambda p: 1.0 / np.linalg.norm(p[k] - p[l], axis=1)
    R_desc = np.stack([desc(pos + rng.normal(0, 0.08, pos.shape)) for _ in range(n_train)], axis=1)      # (Dd, M)
    alpha = rng.normal(0, 2.0e7, (n_train, len(k))) * R_desc.T ** 2
    model = {"sig": np.int64(40), "c": np.float64(-3.2), "std": np.float64(0.07), "z": np.full(n_atoms, 6),
             "R_desc": R_desc, "R_d_desc_alpha": alpha, "perms": np.arange(n_atoms)[None, :],
             "tril_perms_lin": np.arange(len(k))}
    return model, pos


def test_gdml_30_atoms_matches_oracle():
    """config 5's size: N = 30 atoms (3N = 90 > 64, Dd = 435, M = 200): E, grad, Hessian of the HIP kernel against the
    CPU oracle on a synthetic model, then two HK steps through the D > 64 dense path"""
    from oracle import sc_oracle as orc
    from semiclassical_amd.gdml import MolecularGDMLPotential
    from semiclassical_amd import propagators as PR
    torch.set_default_dtype(torch.float64)
    N = 30
    model, pos = synthetic_model(N, 200, 30)
    # energy and forces are linear in the coefficients: scale them to molecular forces (max |dE/dr| = 0.02 Hartree/bohr)
    g0 = orc.GDMLOracle(model).forward(torch.from_numpy(pos.reshape(1, -1)))[1]
    model["R_d_desc_alpha"] = model["R_d_desc_alpha"] * (0.02 / float(g0.abs().max()))
    masses = np.repeat(np.full(N, 12.0 * 1822.888), 3)
    nac0 = np.random.default_rng(1).normal(0, 1e-3, 3 * N)

    class _Fchk(object):
        def nonadiabatic_coupling(self_):
            return nac0

        def masses(self_):
            return masses

        def atomic_numbers(self_):
            return model["z"]
    pot = MolecularGDMLPotential(model, _Fchk())
    ref = orc.GDMLOracle(model)
    rng = np.random.default_rng(5)
    r = torch.from_numpy(pos.reshape(1, -1) + rng.normal(0, 0.05, (6, 3 * N)))            # (B, 3N)
    e_ref, g_ref, h_ref = ref.forward(r)
    v, grad, hess = pot.harmonic_approximation(r.t().contiguous().cuda())
    assert cases.rel_err(cnp(v), e_ref.numpy()) < 1e-10
    assert cases.rel_err(cnp(grad.t()), g_ref.numpy()) < 1e-7
    assert cases.rel_err(cnp(hess.permute(2, 0, 1)), h_ref.numpy()) < 1e-7
    # two HK steps at D = 90 against the oracle propagator on the same surface
    opot = orc.MolecularGDMLOracle(model, masses, nac0, origin=0.0)
    q0 = torch.from_numpy(pos.reshape(-1))
    G = torch.diag(torch.full((3 * N,), 40.0))
    oprop = orc.HKOracle(G, G)
    torch.manual_seed(4)
    oprop.initial_conditions(q0, 0.0 * q0, G, ntraj=12)
    prop = PR.HermanKlukPropagator(G, G, device="cuda")
    prop.set_initial_conditions(q0, 0.0 * q0, G, oprop.zi, oprop.probi)
    for _ in range(2):
        oprop.step(opot, 5.0)
        prop.step(pot, 5.0)
    assert cases.rel_err(cnp(prop.y), oprop.y.numpy()) < 1e-6
    assert cases.rel_err(cnp(prop._c2), oprop.c2.numpy()) < 1e-6
