"""The opt-in separable shortcut: diagonal-state kernel vs the reference goldens and the dense path.

The shortcut carries only the diagonals of the monodromy blocks; c2 is the product of the diagonal of the prefactor
matrix instead of an elimination, so it may differ from the dense kernels by the rounding of D multiplications.
"""
import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu
torch.set_default_dtype(torch.float64)

TOL = 1e-9


def cnp(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("name", ["hk_1d", "hk_as5_chi000", "hk_as5_chi002", "hk_as60", "hk_as60_dt20"])
def test_shortcut_matches_reference_golden(name):
    from tests.engine_cases import engine_potential, engine_propagator
    g = cases.load(name)
    pot = engine_potential(g)
    prop = engine_propagator(g, exploit_separability=True)
    nt, dt, E0 = int(g["nt"]), float(g["dt"]), float(g["E0"])
    cauto, kic = np.zeros(nt, dtype=complex), np.zeros(nt, dtype=complex)
    for t in range(nt):
        assert cases.rel_err(cnp(prop._c2), g["c2"][t]) < TOL, f"c2 at step {t}"
        cauto[t] = prop.autocorrelation(E0)
        kic[t] = prop.ic_correlation(pot, E0)
        prop.step(pot, dt)
        assert prop._mono_stale and prop._mono_is_diag          # the diagonal kernel really ran
        step = t + 1
        if step in g["snaps"]:
            assert np.array_equal(cnp(prop._sgn), g[f"signs_{step}"].real), f"signs at step {step}"
            if f"y_{step}" in g:
                y = cnp(prop.y)                                  # dense blocks rebuilt on demand
                assert cases.rel_err(y, g[f"y_{step}"]) < TOL, f"y at step {step}"
                assert prop._mono_is_diag and not prop._mono_stale
    prop.synchronize()
    assert cases.rel_err(cauto, g["cauto"]) < TOL
    assert cases.rel_err(kic, g["kic"]) < TOL


def test_shortcut_equals_dense_path_and_hands_over():
    """same trajectories through the dense-state kernel and the shortcut; then a dense monodromy set by hand must
    switch the shortcut off"""
    from tests.engine_cases import engine_potential, engine_propagator
    g = cases.load("hk_as60")
    pot = engine_potential(g)
    a, b = engine_propagator(g), engine_propagator(g, exploit_separability=True)
    for _ in range(7):
        a.step(pot, float(g["dt"]))
        b.step(pot, float(g["dt"]))
    assert not a._mono_stale and b._mono_stale
    assert cases.rel_err(cnp(b._c2), cnp(a._c2)) < 1e-12
    assert torch.equal(a._sgn, b._sgn) and torch.equal(a._qp, b._qp) and torch.equal(a._act, b._act)
    assert cases.rel_err(cnp(b.y), cnp(a.y)) < 1e-13
    # a dense (non-diagonal) monodromy: the dense kernel must take over and agree with propagator `a`
    y = a.y.clone()
    d = a.dim
    y[2 * d + 1] += 1e-3                                       # Mqq[0][1] of every trajectory
    a.y = y
    b.y = y
    assert not b._mono_is_diag
    a.step(pot, float(g["dt"]))
    b.step(pot, float(g["dt"]))
    assert not b._mono_stale
    assert torch.equal(a.y, b.y) and torch.equal(a._c2, b._c2)


def test_shortcut_is_refused_where_it_does_not_apply():
    """dense Hessian (methylium) -> the opt-in flag is ignored, results equal the golden"""
    from tests.engine_cases import engine_potential, engine_propagator
    g = cases.load("hk_methylium")
    pot = engine_potential(g)
    prop = engine_propagator(g, exploit_separability=True)
    cauto, kic = prop.run(pot, float(g["dt"]), int(g["nt"]), float(g["E0"]))
    assert not prop._mono_stale and not prop._mono_is_diag
    assert cases.rel_err(cauto, g["cauto"]) < TOL and cases.rel_err(kic, g["kic"]) < TOL
