"""Parity of the HIP Herman-Kluk path (through the C-ABI) with the reference's golden vectors and the oracle.

north_star tolerance: correlation functions within 1e-6 relative (fp64).  The assertions below are tighter
(1e-9) because the engine performs the same arithmetic up to re-association.
"""
import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu

TOL = 1e-9          # what we assert
NORTH_STAR = 1e-6   # what the task requires


def cnp(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("name", cases.HK_CASES)
def test_hk_matches_reference_golden(name):
    from tests.engine_cases import engine_potential, engine_propagator
    g = cases.load(name)
    pot = engine_potential(g)
    prop = engine_propagator(g)
    nt, dt, E0 = int(g["nt"]), float(g["dt"]), float(g["E0"])
    cauto = np.zeros(nt, dtype=complex)
    kic = np.zeros(nt, dtype=complex)
    for t in range(nt):
        assert cases.rel_err(cnp(prop._c2), g["c2"][t]) < TOL, f"c2 at step {t}"
        cauto[t] = prop.autocorrelation(E0)
        kic[t] = prop.ic_correlation(pot, E0)
        prop.step(pot, dt)
        step = t + 1
        if step in g["snaps"]:
            y = cnp(prop.y)
            if f"y_{step}" in g:
                assert cases.rel_err(y, g[f"y_{step}"]) < TOL, f"y at step {step}"
            else:
                d = prop.dim
                assert cases.rel_err(np.vstack((y[:2 * d], y[-1:])), g[f"qpS_{step}"]) < TOL
                assert cases.rel_err(y[:, 0], g[f"ytraj0_{step}"]) < TOL
            assert np.array_equal(cnp(prop._sgn), g[f"signs_{step}"].real), f"signs at step {step}"
            assert cases.rel_err(cnp(prop.autocorrelation_qp()), g[f"cauto_qp_{step}"]) < TOL
    prop.synchronize()
    assert cases.rel_err(cauto, g["cauto"]) < TOL
    assert cases.rel_err(kic, g["kic"]) < TOL
    assert cases.rel_err(cauto, g["cauto"]) < NORTH_STAR and cases.rel_err(kic, g["kic"]) < NORTH_STAR


@pytest.mark.parametrize("name", ["hk_as5_chi002", "hk_methylium", "hk_as60_dt20", "hk_as60_n96", "hk_as33"])
def test_fused_run_equals_stepwise_api(name):
    """run() (no host sync inside the loop) returns what the reference loop returns"""
    from tests.engine_cases import engine_potential, engine_propagator
    g = cases.load(name)
    pot = engine_potential(g)
    prop = engine_propagator(g)
    cauto, kic = prop.run(pot, float(g["dt"]), int(g["nt"]), float(g["E0"]))
    assert cases.rel_err(cauto, g["cauto"]) < TOL
    assert cases.rel_err(kic, g["kic"]) < TOL


def test_state_layout_roundtrip():
    """y (rows, n) -> engine layout -> y is the identity"""
    from tests.engine_cases import engine_propagator
    g = cases.load("hk_as5_chi002")
    prop = engine_propagator(g)
    y = torch.from_numpy(g["y_10"]).cuda()
    prop.y = y
    assert torch.equal(prop.y, y)
    q, p = prop.current_positions_and_momenta()
    assert torch.equal(q, y[:prop.dim]) and torch.equal(p, y[prop.dim:2 * prop.dim])
    Mqq = prop.monodromy_matrices()[0]
    d = prop.dim
    assert torch.equal(Mqq.reshape(d * d, -1), y[2 * d:2 * d + d * d])


def test_hk_matches_oracle_fresh_inputs():
    """seeded inputs that are NOT in the golden set: D=7 anharmonic AS, n=300, against the CPU oracle"""
    from oracle import sc_oracle as orc
    from semiclassical_amd import potentials as P, propagators as PR
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(7)
    D, n, nt = 7, 300, 30
    omega = torch.from_numpy(np.sort(rng.uniform(500, 3000, D)) / 219474.63)
    S = torch.from_numpy(rng.uniform(0.05, 0.4, D) * rng.choice([-1, 1], D))
    nac = torch.from_numpy(rng.normal(0, 1e-3, D))
    chi = torch.full((D,), 0.015, dtype=torch.float64)
    q0 = torch.sqrt(2 * abs(S) / omega) * torch.sign(S)
    p0 = 0.0 * q0
    G = torch.diag(omega)
    E0 = float(0.5 * omega.sum())
    dt = 2.0
    ref = orc.HKOracle(G, G)
    torch.manual_seed(3)
    ref.initial_conditions(q0, p0, G, ntraj=n)
    rpot = orc.MorseOracle(omega, chi, nac)
    rc, rk = orc.run_loop(ref, rpot, dt, nt, E0)
    prop = PR.HermanKlukPropagator(G, G, device="cuda")
    prop.set_initial_conditions(q0, p0, G, ref.zi, ref.probi)
    c, k = prop.run(P.MorsePotential(omega, chi.clone(), nac), dt, nt, E0)
    assert cases.rel_err(c, rc) < TOL and cases.rel_err(k, rk) < TOL


@pytest.mark.parametrize("zero_modes", [0, 2])
def test_dense_and_singular_width_matrices_vs_oracle(zero_modes):
    """separable potential, but DENSE width matrices (rotated, optionally rank deficient as in the reference's
    singular-Gamma tests, tests/test_propagators.py:73-113): the general LDS kernel, the projected prefactor and the
    dense-matrix overlap / NAC kernels against the CPU oracle on fresh inputs; HK and WM"""
    from oracle import sc_oracle as orc
    from semiclassical_amd import potentials as P, propagators as PR
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(11 + zero_modes)
    D, n, nt, dt = 6, 200, 12, 3.0
    omega = torch.from_numpy(np.sort(rng.uniform(600, 2500, D)) / 219474.63)
    S = torch.from_numpy(rng.uniform(0.05, 0.3, D) * rng.choice([-1, 1], D))
    nac = torch.from_numpy(rng.normal(0, 1e-3, D))
    chi = torch.full((D,), 0.01)
    q0 = torch.sqrt(2 * abs(S) / omega) * torch.sign(S)
    p0 = 0.0 * q0
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    w = omega.numpy() * rng.uniform(0.7, 1.4, D)
    w[:zero_modes] = 0.0                                         # zero modes of the width matrices
    G = torch.from_numpy(Q @ np.diag(w) @ Q.T)
    G = 0.5 * (G + G.T)
    E0 = float(0.5 * omega.sum())
    for kind in ("HK", "WM"):
        if kind == "HK":
            ref, prop = orc.HKOracle(G, G), PR.HermanKlukPropagator(G, G, device="cuda")
        else:
            ref, prop = orc.WMOracle(G, G, 50.0, 50.0), PR.WaltonManolopoulosPropagator(G, G, 50.0, 50.0, device="cuda")
        torch.manual_seed(5)
        ref.initial_conditions(q0, p0, G, ntraj=n)
        rc, rk = orc.run_loop(ref, orc.MorseOracle(omega, chi.clone(), nac), dt, nt, E0)
        prop.set_initial_conditions(q0, p0, G, ref.zi, ref.probi)
        assert prop._pre.dprime == D - zero_modes and not prop._pre.diag
        c, k = prop.run(P.MorsePotential(omega, chi.clone(), nac), dt, nt, E0)
        assert cases.rel_err(c, rc) < 1e-8 and cases.rel_err(k, rk) < 1e-8, kind


def test_energy_guard_raises_reference_error():
    """a time step far too large violates <T+V> conservation: same RuntimeError text as the reference"""
    from tests.engine_cases import engine_potential, engine_propagator
    g = cases.load("hk_as5_chi002")
    pot = engine_potential(g)
    prop = engine_propagator(g)
    with pytest.raises(RuntimeError, match="average energy of classical trajectories is not conserved"):
        for _ in range(6):
            prop.step(pot, 400.0 * float(g["dt"]))
        prop.synchronize()


def test_wrong_dimension_asserts():
    from tests.engine_cases import engine_potential, engine_propagator
    prop = engine_propagator(cases.load("hk_as5_chi002"))
    pot = engine_potential(cases.load("hk_1d"))
    with pytest.raises(AssertionError, match="potential has wrong dimensions"):
        prop.step(pot, 0.1)


def _prefactor_of_state(name, make_blocks, without_flags=False):
    """c2 of the HIP engine and of the oracle for a hand-made monodromy state"""
    from tests.engine_cases import engine_propagator
    g = cases.load(name)
    prop = engine_propagator(g)
    if without_flags:
        prop._state.flags = None            # a C-ABI caller that passes no flag array (legal: include/semiclassical_hip.h)
    ref = cases.oracle_propagator(g)
    d, n = prop.dim, prop.ntraj
    y = ref.y.clone()
    blocks = make_blocks(d, n)
    for k, blk in enumerate(blocks):
        y[2 * d + k * d * d: 2 * d + (k + 1) * d * d] = blk.reshape(d * d, n)
    ref.y = y
    ref._prefactor()
    prop.y = y.cuda()
    prop._prefactor_initial()
    torch.cuda.synchronize()
    return cnp(prop._c2), ref.c2.numpy(), prop


def test_dense_monodromy_through_fast_path():
    """random dense (non-diagonal) monodromy blocks: the register elimination really has to pivot"""
    torch.manual_seed(5)

    def blocks(d, n):
        eye = torch.eye(d).unsqueeze(2)
        return [eye + 0.3 * torch.randn(d, d, n), 0.3 * torch.randn(d, d, n),
                0.3 * torch.randn(d, d, n), eye + 0.3 * torch.randn(d, d, n)]
    for name in ("hk_as60", "hk_as5_chi002"):
        got, want, prop = _prefactor_of_state(name, blocks)
        assert np.max(np.abs(got - want) / np.abs(want)) < 1e-10
        assert int(prop._flags[:-2].sum().item()) == 0         # flags of handed-over trajectories are cleared again
        assert int(prop._flags[-2].item()) < prop.ntraj        # the register elimination kept some of them


def test_weak_pivot_fallback():
    """monodromy blocks = cyclic shift by 20 columns: every in-block pivot candidate is zero, so the fast path
    must hand the trajectory to the fully pivoted elimination (and clear its flag afterwards)"""
    def blocks(d, n):
        shift = torch.roll(torch.eye(d), 20, dims=1).unsqueeze(2).expand(-1, -1, n).clone()
        shift = shift * (1.0 + 0.1 * torch.rand(d, d, n))
        zero = torch.zeros(d, d, n)
        return [shift, zero, zero, shift.clone()]
    got, want, prop = _prefactor_of_state("hk_as60", blocks)
    assert np.all(np.abs(want) > 0)
    assert np.max(np.abs(got - want) / np.abs(want)) < 1e-10
    assert int(prop._flags[:-2].sum().item()) == 0             # flags cleared by the fix-up pass ...
    assert int(prop._flags[-2].item()) == prop.ntraj           # ... which was needed for every trajectory


@pytest.mark.parametrize("name", ["hk_as60", "hk_as5_chi002", "hk_methylium"])
def test_zero_leading_pivots_without_a_flag_array(name):
    """sc_state.flags == NULL: there is no fix-up launch, so the fixed-pivot-order register kernels must not be used at all
    (a zero leading pivot would end as inf / NaN in c2).  Cyclically shifted monodromy blocks make every leading pivot
    zero; the call has to take the fully pivoted kernel and return the oracle's determinants."""
    def blocks(d, n):
        gen = torch.Generator().manual_seed(3)
        shift = torch.roll(torch.eye(d), max(1, d // 3), dims=1).unsqueeze(2).expand(-1, -1, n).clone()
        shift = shift * (1.0 + 0.1 * torch.rand(d, d, n, generator=gen))
        zero = torch.zeros(d, d, n)
        return [shift, zero, zero, shift.clone()]
    got, want, prop = _prefactor_of_state(name, blocks, without_flags=True)
    assert np.all(np.isfinite(got)) and np.all(np.abs(want) > 0)
    assert np.max(np.abs(got - want) / np.abs(want)) < 1e-10


def test_singular_prefactor_matrix_gives_zero_determinant():
    """a zero row in every monodromy block (row 7) makes the prefactor matrix singular for every second trajectory: the
    register elimination meets a zero pivot (flag bit in LDS, no determinant carried in registers) and must return
    c2 = 0 exactly -- as torch.det of the oracle does up to rounding -- while the other trajectories are untouched"""
    def blocks(d, n):
        gen = torch.Generator().manual_seed(8)
        out = [torch.eye(d).unsqueeze(2).expand(-1, -1, n).clone() + 0.1 * torch.randn(d, d, n, generator=gen) for _ in range(4)]
        for blk in out:
            blk[7, :, ::2] = 0.0
        return out
    got, want, prop = _prefactor_of_state("hk_as60", blocks)
    assert np.all(got[::2] == 0.0)
    assert np.max(np.abs(want[::2])) < 1e-10 * np.max(np.abs(want[1::2]))
    assert np.max(np.abs(got[1::2] - want[1::2]) / np.abs(want[1::2])) < 1e-9


def test_unsupported_sizes_fail_loudly():
    """no silent fallback: what the kernels cannot hold is refused with the C-ABI's error text"""
    from semiclassical_amd import propagators as PR
    from semiclassical_amd._lib import EngineError, lib, check, sc_state, sc_hk_consts, sc_potential, SC_POT_MORSE
    torch.set_default_dtype(torch.float64)
    D = 70
    G = torch.diag(torch.linspace(0.5, 1.5, D))
    q0 = torch.zeros(D)
    # WM without the scratch block its matrices need at this size is refused (the propagator allocates it itself)
    wm = PR.WaltonManolopoulosPropagator(G, G, 10.0, 10.0, device="cuda")
    wm.initial_conditions(q0, q0, G, ntraj=8)
    assert wm._wm_scratch is not None and abs(wm.autocorrelation() - 1.0) < 0.5
    wm._wm.scratch, wm._wm.scratch_bytes = None, 0
    with pytest.raises(EngineError, match="scratch"):
        wm._wm_launch(0)
    # the fused step entry point is limited to D <= 64
    st = sc_state(n=1, dim=D)
    with pytest.raises(EngineError, match="outside 1..64"):
        check(lib.sc_hk_step(sc_potential(kind=SC_POT_MORSE, dim=D), st, sc_hk_consts(dim=D, dprime=D, diag=1), 0.1, 0,
                             None, None))
    with pytest.raises(EngineError, match="null argument"):
        check(lib.sc_hk_step(None, None, None, 0.1, 0, None, None))
    # beyond D = 96 the dense path needs its scratch block
    assert lib.sc_dense_mono_scratch_bytes(10, 100, 100) > 0 and lib.sc_dense_mono_scratch_bytes(10, 64, 64) == 0
    with pytest.raises(EngineError, match="mono_sums scratch"):
        check(lib.sc_dense_mono_step(sc_state(n=1, dim=100), sc_hk_consts(dim=100, dprime=100, diag=1), None, None, None,
                                     0.0, 1, None))


@pytest.mark.parametrize("n", [1, 3, 65])
def test_ragged_trajectory_counts(n):
    """a single trajectory / counts that fill no tile or wavefront: HK (dense state and shortcut) and WM against the
    oracle on the first n golden initial conditions"""
    from tests.engine_cases import engine_potential
    from semiclassical_amd import propagators as PR
    from oracle import sc_oracle as orc
    torch.set_default_dtype(torch.float64)            # the oracle follows the reference's global default (cli.py:121)
    g = cases.load("hk_as5_chi002")
    pot, opot = engine_potential(g), cases.oracle_potential(g)
    Gi = cases.T(g["Gamma_i"])
    zi, probi = cases.T(g["zi"])[:, :n].contiguous(), cases.T(g["probi"])[:n].contiguous()
    for make_ref, make_eng in ((lambda: orc.HKOracle(Gi, Gi), lambda: PR.HermanKlukPropagator(Gi, Gi, device="cuda")),
                               (lambda: orc.HKOracle(Gi, Gi),
                                lambda: PR.HermanKlukPropagator(Gi, Gi, device="cuda", exploit_separability=True)),
                               (lambda: orc.WMOracle(Gi, Gi, 100.0, 100.0),
                                lambda: PR.WaltonManolopoulosPropagator(Gi, Gi, 100.0, 100.0, device="cuda"))):
        ref, prop = make_ref(), make_eng()
        ref.set_initial_conditions(cases.T(g["q0"]), cases.T(g["p0"]), cases.T(g["Gamma_0"]), zi, probi)
        prop.set_initial_conditions(cases.T(g["q0"]), cases.T(g["p0"]), cases.T(g["Gamma_0"]), zi, probi)
        rc, rk = orc.run_loop(ref, opot, float(g["dt"]), 6, float(g["E0"]))
        c, k = prop.run(pot, float(g["dt"]), 6, float(g["E0"]))
        assert cases.rel_err(c, rc) < 1e-8 and cases.rel_err(k, rk) < 1e-8


@pytest.mark.parametrize("D", [2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64])
def test_fast_path_at_tile_boundaries(D):
    """dimensions around the 16-row / 16-column tile edges of the fast kernels (one row in the last block, full blocks,
    ...): anharmonic AS model of D modes, dense-state kernel AND diagonal shortcut against the CPU oracle, plus a dense
    random monodromy pushed through the register elimination"""
    import bench
    from oracle import sc_oracle as orc
    from semiclassical_amd import potentials as P, propagators as PR
    torch.set_default_dtype(torch.float64)
    omega, chi, nac, q0, _ = bench.as60_model(D)
    dt, n, nt = 4.0, 12, 3
    G = torch.diag(omega)
    E0 = float(0.5 * omega.sum())
    ref = orc.HKOracle(G, G)
    torch.manual_seed(D)
    ref.initial_conditions(q0, 0.0 * q0, G, ntraj=n)
    rc, rk = orc.run_loop(ref, orc.MorseOracle(omega, chi.clone(), nac), dt, nt, E0)
    for kw in ({}, {"exploit_separability": True}):
        prop = PR.HermanKlukPropagator(G, G, device="cuda", **kw)
        prop.set_initial_conditions(q0, 0.0 * q0, G, ref.zi, ref.probi)
        c, k = prop.run(P.MorsePotential(omega, chi.clone(), nac), dt, nt, E0)
        assert cases.rel_err(c, rc) < 1e-9 and cases.rel_err(k, rk) < 1e-9
        assert cases.rel_err(cnp(prop.y), ref.y.numpy()) < 1e-10
    # dense random blocks: prefactor of the engine vs the oracle's (pivoting, fallback)
    y = ref.y.clone()
    gen = torch.Generator().manual_seed(100 + D)
    y[2 * D:2 * D + 4 * D * D] += 0.2 * torch.randn((4 * D * D, n), generator=gen)
    ref.y = y
    ref._prefactor()
    prop.y = y.cuda()
    prop._prefactor_initial()
    assert cases.rel_err(cnp(prop._c2), ref.c2.numpy()) < 1e-9


@pytest.mark.parametrize("D", [17, 33, 50, 60, 64])
def test_tiled_monodromy_layout_roundtrip_and_parity(D):
    """SC_MONO_TILED16 (include/semiclassical_hip.h): sc_mono_convert against the documented offset formula, in-place
    round trip, and the fast path giving the same state in either storage order"""
    from semiclassical_amd import _lib, potentials as P, propagators as PR
    from semiclassical_amd._lib import lib, check
    from oracle import sc_oracle as orc
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(D)
    n = 37
    omega = torch.from_numpy(np.sort(rng.uniform(300, 3000, D)) / 219474.63)
    chi = torch.full((D,), 0.02)
    nac = torch.from_numpy(rng.normal(0, 1e-4, D))
    q0 = torch.from_numpy(rng.uniform(-3, 3, D))
    G = torch.diag(omega)

    def offset(p, a, b):
        ra, rb = a // 16, b // 16
        nra, ncb = min(16, D - 16 * ra), min(16, D - 16 * rb)
        return 4 * (16 * ra * D + 16 * nra * rb) + (p // 2) * 2 * nra * ncb + 2 * ((a % 16) * ncb + (b % 16)) + p % 2
    perm = np.array([offset(p, a, b) for p in range(4) for a in range(D) for b in range(D)])
    assert np.array_equal(np.sort(perm), np.arange(4 * D * D))                 # a bijection

    props = []
    for tiled in (True, False):
        prop = PR.HermanKlukPropagator(G, G, device="cuda")
        prop._tiled_fast_path = tiled
        prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, generator=torch.Generator().manual_seed(5))
        # dense random monodromy blocks, so that every element position matters
        y = prop.y
        y[2 * D:2 * D + 4 * D * D] = torch.from_numpy(rng.uniform(-1, 1, (4 * D * D, n))).cuda() + y[2 * D:2 * D + 4 * D * D]
        prop.y = y
        props.append(prop)
    a, b = props
    b.y = a.y
    pot = P.MorsePotential(omega, chi.clone(), nac)
    for _ in range(3):
        a.step(pot, 0.3)
        b.step(pot, 0.3)
    assert a._state.mono_layout == _lib.SC_MONO_TILED16 and b._state.mono_layout == _lib.SC_MONO_ROWMAJOR
    raw_tiled = a._mono.reshape(n, -1).clone()
    assert torch.equal(raw_tiled[:, torch.from_numpy(perm).cuda()], b._mono.reshape(n, -1))     # documented order, same numbers
    assert torch.equal(a._c2, b._c2) and torch.equal(a._sgn, b._sgn)
    ya = a.y                                                                  # converts back in place
    assert a._state.mono_layout == _lib.SC_MONO_ROWMAJOR
    assert torch.equal(ya, b.y)
    check(lib.sc_mono_convert(a._state, _lib.SC_MONO_TILED16, None))
    torch.cuda.synchronize()
    assert torch.equal(a._mono.reshape(n, -1), raw_tiled)
    a._state.mono_layout = _lib.SC_MONO_TILED16
    with pytest.raises(_lib.EngineError, match="tiled monodromy layout"):
        check(lib.sc_state_to_reference(a._state, _lib.ptr(ya), None))


@pytest.mark.parametrize("D,zero_modes,diag", [(12, 6, False), (12, 0, True), (9, 6, False), (9, 0, True),
                                               (6, 0, True), (6, 0, False), (3, 0, True), (5, 0, True),
                                               (6, 5, False), (9, 5, False), (12, 5, False), (15, 6, False), (15, 0, True)])
def test_constant_hessian_register_kernel_vs_oracle(D, zero_modes, diag):
    """dense constant Hessian (MolecularHarmonicPotential) on random couplings: the shapes of the register kernel that
    multiplies with the RK4 step matrix (sc_hk_step_lin.hip), diagonal and projected prefactor, against the oracle's
    staged RK4; (5, 0, True) is not instantiated and checks the hand-over to the general kernel"""
    from oracle import sc_oracle as orc
    from semiclassical_amd import potentials as P, propagators as PR
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(100 * D + zero_modes + diag)
    n, nt, dt = 150, 10, 4.0
    masses = rng.uniform(1800.0, 22000.0, D)
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    om = rng.uniform(500, 3000, D) / 219474.63
    sm = np.sqrt(masses)
    hess0 = (Q * om ** 2) @ Q.T * np.outer(sm, sm)
    hess0 = 0.5 * (hess0 + hess0.T)
    pos0, grad0 = rng.normal(0, 0.1, D), rng.normal(0, 1e-3, D)
    nac0 = rng.normal(0, 1e-2, D)
    args = (pos0, np.float64(-0.3), grad0, hess0, masses, nac0)
    w = om * rng.uniform(0.7, 1.4, D)
    if diag:
        G = np.diag(w * masses)
    else:
        w[:zero_modes] = 0.0
        U, _ = np.linalg.qr(rng.standard_normal((D, D)))
        G = (U * w) @ U.T * np.outer(sm, sm)
        G = 0.5 * (G + G.T)
    G = torch.from_numpy(G)
    q0, p0 = torch.from_numpy(pos0 + rng.normal(0, 0.05, D)), torch.zeros(D)
    ref, prop = orc.HKOracle(G, G), PR.HermanKlukPropagator(G, G, device="cuda")
    torch.manual_seed(3)
    ref.initial_conditions(q0, p0, G, ntraj=n)
    rc, rk = orc.run_loop(ref, orc.MolecularHarmonicOracle(*args, origin=-0.3), dt, nt, 0.01)
    prop.set_initial_conditions(q0, p0, G, ref.zi, ref.probi)
    assert prop._pre.dprime == D - zero_modes and bool(prop._pre.diag) == diag
    pot = P.MolecularHarmonicPotential.from_arrays(*args, origin=-0.3)
    c, k = prop.run(pot, dt, nt, 0.01)
    assert cases.rel_err(c, rc) < TOL and cases.rel_err(k, rk) < TOL
    for a, b in zip(prop.current_positions_and_momenta() + prop.monodromy_matrices(),
                    ref.current_positions_and_momenta() + ref.monodromy_matrices()):
        assert cases.rel_err(a.cpu(), b) < 1e-11
    assert cases.rel_err(prop.classical_action().cpu(), ref.classical_action()) < 1e-11
    assert abs(prop.mean_energy() - float(ref.eom.en_mean)) < 1e-11 * max(1.0, abs(float(ref.eom.en_mean)))


@pytest.mark.parametrize("D,zero_modes,diag", [(6, 0, True), (12, 6, False)])
def test_constant_hessian_register_kernel_weak_pivots(D, zero_modes, diag):
    """the register kernel of sc_hk_step_lin.hip eliminates in a fixed order: a state whose monodromy blocks are cyclic
    shifts puts (almost) zero on the first pivot, so every trajectory is flagged and goes through the pivoted fix-up
    launch; the prefactor must still agree with the oracle's LU"""
    from oracle import sc_oracle as orc
    from semiclassical_amd import potentials as P, propagators as PR
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(7 * D + zero_modes)
    n, dt = 96, 0.5
    masses = rng.uniform(1800.0, 22000.0, D)
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    om = rng.uniform(500, 3000, D) / 219474.63
    sm = np.sqrt(masses)
    hess0 = (Q * om ** 2) @ Q.T * np.outer(sm, sm)
    hess0 = 0.5 * (hess0 + hess0.T)
    pos0, grad0, nac0 = rng.normal(0, 0.1, D), rng.normal(0, 1e-3, D), rng.normal(0, 1e-2, D)
    args = (pos0, np.float64(0.0), grad0, hess0, masses, nac0)
    w = om * rng.uniform(0.7, 1.4, D)
    if diag:
        G = np.diag(w * masses)
    else:
        w[:zero_modes] = 0.0
        U = np.eye(D)                                    # widths diagonal in the mass-weighted frame: L picks coordinate rows
        G = (U * w) @ U.T * np.outer(sm, sm)
    G = torch.from_numpy(G)
    q0, p0 = torch.from_numpy(pos0 + rng.normal(0, 0.05, D)), torch.zeros(D)
    ref, prop = orc.HKOracle(G, G), PR.HermanKlukPropagator(G, G, device="cuda")
    torch.manual_seed(4)
    ref.initial_conditions(q0, p0, G, ntraj=n)
    prop.set_initial_conditions(q0, p0, G, ref.zi, ref.probi)
    assert prop._pre.dprime == D - zero_modes and bool(prop._pre.diag) == diag
    shift = np.roll(np.eye(D), 1, axis=1)
    y = ref.y.clone()
    for blk, mat in enumerate((shift, 0.0 * shift, 0.0 * shift, shift)):          # Mqq, Mqp, Mpq, Mpp
        lo = 2 * D + blk * D * D
        y[lo:lo + D * D] = torch.from_numpy(mat.reshape(-1, 1) + 1e-3 * rng.standard_normal((D * D, n)))
    ref.y = y.clone()
    prop.y = y.cuda()
    pot = P.MolecularHarmonicPotential.from_arrays(*args)
    ref_pot = orc.MolecularHarmonicOracle(*args)
    flagged = 0
    for _ in range(2):
        ref.step(ref_pot, dt)
        prop.step(pot, dt)
        flagged += int(prop._flags[-2].item())
        assert int(prop._flags[:-2].sum().item()) == 0
        got, want = prop.semiclassical_prefactor().cpu().numpy(), ref.semiclassical_prefactor().numpy()
        assert np.max(np.abs(got - want) / np.abs(want)) < 1e-9
    assert flagged > 0, "no trajectory took the fix-up launch: the test no longer exercises it"


@pytest.mark.parametrize("D,n", [(6, 200), (12, 17), (12, 2 * 8192 + 37)])
def test_constant_hessian_register_kernel_unaligned_state_takes_general_kernel(D, n):
    """the row prefetch of sc_hk_step_lin.hip moves 16-byte units: a caller of the C-ABI whose state arrays are only 8-byte
    aligned must still get the right answer (the dispatcher hands such a state to the general kernel).  The large batch
    gives every persistent workgroup several passes (rows requested one pass ahead) and a ragged last one."""
    from semiclassical_amd import potentials as P, propagators as PR
    from semiclassical_amd._lib import lib, check, ptr
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(11)
    dt = 3.0
    masses = rng.uniform(1800.0, 22000.0, D)
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    om = rng.uniform(500, 3000, D) / 219474.63
    sm = np.sqrt(masses)
    hess0 = (Q * om ** 2) @ Q.T * np.outer(sm, sm)
    hess0 = 0.5 * (hess0 + hess0.T)
    pos0 = rng.normal(0, 0.1, D)
    args = (pos0, np.float64(0.0), rng.normal(0, 1e-3, D), hess0, masses, rng.normal(0, 1e-2, D))
    G = torch.from_numpy(np.diag(om * masses))
    q0 = torch.from_numpy(pos0 + rng.normal(0, 0.05, D))
    pot = P.MolecularHarmonicPotential.from_arrays(*args)
    a, b = (PR.HermanKlukPropagator(G, G, device="cuda") for _ in range(2))
    for prop in (a, b):
        prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, generator=torch.Generator().manual_seed(2))
    for _ in range(3):
        a.step(pot, dt)
    desc = b._potential_descriptor(pot, dt)
    st = type(b._state).from_buffer_copy(b._state)
    shifted = torch.zeros(n + 1, device="cuda")
    act = shifted[1:]                                 # 8 bytes off a 16-byte boundary
    act.copy_(b._act)
    assert act.data_ptr() % 16 == 8
    st.act = ptr(act)
    for _ in range(3):
        check(lib.sc_hk_step(desc, st, b._hk, dt, 0, ptr(b._epart), b._stream()))
    torch.cuda.synchronize()
    for x, y in ((a._qp, b._qp), (a._act, act), (a._mono, b._mono)):
        assert cases.rel_err(x.cpu(), y.cpu()) < 1e-12
    assert cases.rel_err(torch.view_as_real(a._c2).cpu(), torch.view_as_real(b._c2).cpu()) < 1e-11
    assert torch.equal(a._sgn, b._sgn)


@pytest.mark.parametrize("D,n", [(60, 2500), (33, 1500)])
def test_trajectory_cursor_and_the_flagless_path_agree(D, n):
    """more trajectories than persistent workgroups (1024): the fast kernel hands them out through the device-side cursor
    (sc_state.flags[n + 1]) and must process every trajectory exactly once.  A C-ABI caller that passes sc_state.flags =
    NULL has no fix-up launch behind the register elimination, so sc_hk_step gives it the fully pivoted LDS kernel for every
    trajectory (round 4; before that: the register kernel with a static stride and unchecked pivots): same state bit for
    bit (the RK4 arithmetic of a row does not depend on the kernel), determinants to rounding, signs exact -- and both equal
    to the oracle on a sample of trajectories"""
    import bench
    from oracle import sc_oracle as orc
    from semiclassical_amd import _lib, potentials as P, propagators as PR
    from semiclassical_amd._lib import lib, check, ptr
    torch.set_default_dtype(torch.float64)
    omega, chi, nac, q0, _ = bench.as60_model(D)
    G, dt = torch.diag(omega), 4.0
    pot = P.MorsePotential(omega, chi.clone(), nac)
    props = []
    for _ in range(2):
        prop = PR.HermanKlukPropagator(G, G, device="cuda")
        prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, generator=torch.Generator().manual_seed(D))
        props.append(prop)
    a, b = props
    for _ in range(3):
        a.step(pot, dt)
    # b: the same steps through the C-ABI with flags = NULL (row-major state: the general kernel does not take the tiled order)
    desc = b._potential_descriptor(pot, dt)
    st = type(b._state).from_buffer_copy(b._state)
    st.flags = None
    for _ in range(3):
        check(lib.sc_hk_step(desc, st, b._hk, dt, 0, ptr(b._epart), b._stream()))
    torch.cuda.synchronize()
    assert int(a._flags[-1].item()) == n                        # one draw from the cursor per processed trajectory
    a._set_mono_layout(_lib.SC_MONO_ROWMAJOR)
    for x, y in ((a._qp, b._qp), (a._act, b._act), (a._mono, b._mono)):
        assert cases.rel_err(x.cpu(), y.cpu()) < 1e-13
    assert cases.rel_err(torch.view_as_real(a._c2).cpu(), torch.view_as_real(b._c2).cpu()) < 1e-11
    assert torch.equal(a._sgn, b._sgn)
    # oracle on the first and the last 8 trajectories
    pick = torch.cat((torch.arange(8), torch.arange(n - 8, n)))
    ref = orc.HKOracle(G, G)
    ref.set_initial_conditions(q0, 0.0 * q0, G, a.zi.cpu()[:, pick], a.probi.cpu()[pick])
    opot = orc.MorseOracle(omega, chi.clone(), nac)
    for _ in range(3):
        ref.step(opot, dt)
    assert cases.rel_err(cnp(a.y)[:, pick.numpy()], ref.y.numpy()) < 1e-10
    assert cases.rel_err(cnp(a._c2)[pick.numpy()], ref.c2.numpy()) < 1e-9


def test_trajectory_cursor_under_graph_replay():
    """the cursor is zeroed by a memset inside sc_hk_step: captured into the HIP graph of run(use_graph=True), it must be
    reset on every replay (n > 1024 trajectories, D = 33: eager loop and graph replay give identical results)"""
    import bench
    from semiclassical_amd import potentials as P, propagators as PR
    torch.set_default_dtype(torch.float64)
    D, n, nt = 33, 2200, 7
    omega, chi, nac, q0, _ = bench.as60_model(D)
    G, dt, E0 = torch.diag(omega), 4.0, float(0.5 * omega.sum())
    out = []
    for use_graph in (False, True):
        prop = PR.HermanKlukPropagator(G, G, device="cuda")
        prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, generator=torch.Generator().manual_seed(3))
        c, k = prop.run(P.MorsePotential(omega, chi.clone(), nac), dt, nt, E0, use_graph=use_graph)
        out.append((c, k, prop._c2.clone(), prop._qp.clone()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert torch.equal(out[0][2], out[1][2]) and torch.equal(out[0][3], out[1][3])


@pytest.mark.parametrize("name", ["hk_as5_chi002", "hk_as5_chi000", "hk_1d"])
def test_whole_loop_kernel_matches_stepwise_path_and_golden(name):
    """sc_hk_run (the caller loop of cli.py:401-436 as ONE launch; separable potential, diagonal widths, D <= 12) against the
    step-by-step launches of the same engine and against the reference's golden C(t), k_ic(t): same per-trajectory
    arithmetic, so the state agrees to rounding of the summation order (asserted 1e-13), signs bit-exact."""
    from tests.engine_cases import engine_potential, engine_propagator
    g = cases.load(name)
    nt, dt, E0 = int(g["nt"]), float(g["dt"]), float(g["E0"])
    fused, stepwise = engine_propagator(g), engine_propagator(g)
    stepwise._whole_loop_ok = False
    pot = engine_potential(g)
    assert fused._whole_loop_applies(fused._potential_descriptor(pot, dt))
    c1, k1 = fused.run(pot, dt, nt, E0)
    c2, k2 = stepwise.run(pot, dt, nt, E0)
    assert cases.rel_err(c1, c2) < 1e-13 and cases.rel_err(k1, k2) < 1e-13
    assert cases.rel_err(c1, g["cauto"]) < 1e-9 and cases.rel_err(k1, g["kic"]) < 1e-9
    assert torch.equal(fused._sgn, stepwise._sgn)
    assert cases.rel_err(cnp(fused.y), cnp(stepwise.y)) < 1e-13
    assert cases.rel_err(cnp(fused._c2), cnp(stepwise._c2)) < 1e-12
    assert abs(fused.t - stepwise.t) == 0.0 and fused._nsteps == stepwise._nsteps
    assert cases.rel_err(cnp(fused._elog), cnp(stepwise._elog)) < 1e-12          # the energy guard saw the same means
    # and the loop can be continued step by step from the state the kernel left
    fused.step(pot, dt); stepwise.step(pot, dt)
    assert abs(fused.autocorrelation(E0) - stepwise.autocorrelation(E0)) < 1e-13 * abs(stepwise.autocorrelation(E0))


def test_whole_loop_kernel_long_run_goes_in_chunks():
    """more steps than one launch takes (4096: the per-step partial sums of every wavefront are bounded): the launches of a
    long run() continue each other on the device and fill consecutive slots; against the step-by-step path"""
    from tests.engine_cases import engine_potential, engine_propagator
    g = cases.load("hk_1d")
    nt, dt, E0 = 4096 + 37, float(g["dt"]) * 0.02, float(g["E0"])
    fused, stepwise = engine_propagator(g), engine_propagator(g)
    stepwise._whole_loop_ok = False
    pot = engine_potential(g)
    c1, k1 = fused.run(pot, dt, nt, E0)
    c2, k2 = stepwise.run(pot, dt, nt, E0)
    assert c1.shape == (nt,) and cases.rel_err(c1, c2) < 1e-12 and cases.rel_err(k1, k2) < 1e-12
    assert torch.equal(fused._sgn, stepwise._sgn) and cases.rel_err(cnp(fused.y), cnp(stepwise.y)) < 1e-12
    assert fused._nsteps == stepwise._nsteps == nt and abs(fused.t - stepwise.t) == 0.0
    assert cases.rel_err(cnp(fused._elog), cnp(stepwise._elog)) < 1e-11


def test_whole_loop_kernel_with_dense_blocks_and_ragged_batch():
    """random dense monodromy blocks (the fixed-order elimination meets weak pivots: in-kernel pivoted repeat), D = 12, and
    a batch that is not a multiple of 16"""
    from semiclassical_amd import potentials as P, propagators as PR
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(3)
    D, n, nt, dt = 12, 203, 6, 2.0
    omega = torch.from_numpy(np.sort(rng.uniform(500, 3000, D)) / 219474.63)
    S = torch.from_numpy(rng.uniform(0.05, 0.4, D))
    nac = torch.from_numpy(rng.normal(0, 1e-3, D))
    G = torch.diag(omega)
    q0, E0 = torch.sqrt(2 * S / omega), float(0.5 * omega.sum())
    props = []
    gen = torch.Generator().manual_seed(2)
    blocks = [torch.eye(D).unsqueeze(2) * (k in (0, 3)) + 0.4 * torch.randn(D, D, n, generator=gen) for k in range(4)]
    for whole in (True, False):
        prop = PR.HermanKlukPropagator(G, G, device="cuda")
        prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, generator=torch.Generator().manual_seed(5))
        y = prop.y
        for k, blk in enumerate(blocks):
            y[2 * D + k * D * D: 2 * D + (k + 1) * D * D] = blk.reshape(D * D, n).cuda()
        prop.y = y
        prop._whole_loop_ok = whole
        props.append((prop, prop.run(P.MorsePotential(omega, torch.full((D,), 0.02), nac), dt, nt, E0)))
    (a, (ca, ka)), (b, (cb, kb)) = props
    assert cases.rel_err(ca, cb) < 1e-12 and cases.rel_err(ka, kb) < 1e-12
    assert torch.equal(a._sgn, b._sgn)
    assert cases.rel_err(cnp(a._c2), cnp(b._c2)) < 1e-11 and cases.rel_err(cnp(a.y), cnp(b.y)) < 1e-13


def test_whole_loop_kernel_for_the_constant_hessian_matches_stepwise_path_and_golden():
    """sc_hk_run for a constant dense Hessian (round 4; hk_run_lin_kernel: methylium, D = 12, rank-6 Cartesian widths): the
    reference's golden C(t), k_ic(t) through run() at 1e-9, the step-at-a-time path of the same engine at 1e-13 with bit-equal
    branch signs, the same state, determinants and energy means -- and the loop continues step by step afterwards."""
    from tests.engine_cases import engine_potential, engine_propagator
    g = cases.load("hk_methylium")
    nt, dt, E0 = int(g["nt"]), float(g["dt"]), float(g["E0"])
    fused, stepwise = engine_propagator(g), engine_propagator(g)
    stepwise._whole_loop_ok = False
    pot = engine_potential(g)
    assert fused._whole_loop_applies(fused._potential_descriptor(pot, dt))
    c1, k1 = fused.run(pot, dt, nt, E0)
    c2, k2 = stepwise.run(pot, dt, nt, E0)
    assert cases.rel_err(c1, c2) < 1e-13 and cases.rel_err(k1, k2) < 1e-13, (cases.rel_err(c1, c2), cases.rel_err(k1, k2))
    assert cases.rel_err(c1, g["cauto"]) < 1e-9 and cases.rel_err(k1, g["kic"]) < 1e-9
    assert torch.equal(fused._sgn, stepwise._sgn)
    assert cases.rel_err(cnp(fused.y), cnp(stepwise.y)) < 1e-13
    assert cases.rel_err(cnp(fused._c2), cnp(stepwise._c2)) < 1e-12
    assert abs(fused.t - stepwise.t) == 0.0 and fused._nsteps == stepwise._nsteps
    assert cases.rel_err(cnp(fused._elog), cnp(stepwise._elog)) < 1e-12
    fused.step(pot, dt); stepwise.step(pot, dt)
    assert abs(fused.autocorrelation(E0) - stepwise.autocorrelation(E0)) < 1e-13 * abs(stepwise.autocorrelation(E0))


@pytest.mark.parametrize("D,zero_modes,diag", [(12, 6, False), (12, 0, True), (9, 6, False), (6, 0, True), (6, 0, False), (3, 0, True),
                                               (6, 5, False), (9, 5, False), (12, 5, False), (9, 0, True)])
def test_whole_loop_constant_hessian_shapes_dense_blocks_and_ragged_batch(D, zero_modes, diag):
    """every instantiated shape of hk_run_lin_kernel (diagonal widths, rotated full-rank and rank-deficient widths) on a ragged
    batch with random dense monodromy blocks -- part of the fixed-order eliminations meet weak pivots and are repeated with
    the pivot searched among the lanes -- against the step-at-a-time path (whose weak pivots go to the fix-up launch)"""
    from semiclassical_amd import potentials as P, propagators as PR
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(100 + 7 * D + zero_modes)
    n, nt, dt = 203, 5, 2.0
    w = np.sort(rng.uniform(500, 3000, D)) / 219474.63
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    masses = rng.uniform(0.8, 1.5, D) * 1822.0
    hess = (Q * (w ** 2 * 1822.0)) @ Q.T
    hess = 0.5 * (hess + hess.T)
    if diag:
        G = torch.diag(torch.from_numpy(w * 1822.0))
    else:
        ww = w * 1822.0 * rng.uniform(0.7, 1.4, D)
        ww[:zero_modes] = 0.0
        Qg, _ = np.linalg.qr(rng.standard_normal((D, D)))
        G = torch.from_numpy((Qg * ww) @ Qg.T)
        G = 0.5 * (G + G.T)
    pos0 = rng.normal(0, 0.05, D)
    pot_args = (pos0, -0.3, rng.normal(0, 1e-3, D), hess, masses, rng.normal(0, 1e-3, D))
    q0 = torch.from_numpy(pos0 + rng.normal(0, 0.02, D))
    E0 = 0.01
    gen = torch.Generator().manual_seed(2)
    blocks = [torch.eye(D).unsqueeze(2) * (k in (0, 3)) + 0.4 * torch.randn(D, D, n, generator=gen) for k in range(4)]
    props = []
    for whole in (True, False):
        pot = P.MolecularHarmonicPotential.from_arrays(*pot_args, origin=-0.3)
        prop = PR.HermanKlukPropagator(G, G, device="cuda")
        prop.initial_conditions(q0, 0.0 * q0, G, ntraj=n, generator=torch.Generator().manual_seed(5))
        y = prop.y
        for k, blk in enumerate(blocks):
            y[2 * D + k * D * D: 2 * D + (k + 1) * D * D] = blk.reshape(D * D, n).cuda()
        prop.y = y
        prop._whole_loop_ok = whole
        if whole:
            assert prop._whole_loop_applies(prop._potential_descriptor(pot, dt)), "shape not taken by the whole-loop kernel"
        props.append((prop, prop.run(pot, dt, nt, E0)))
    (a, (ca, ka)), (b, (cb, kb)) = props
    assert np.isfinite(ca).all() and cases.rel_err(ca, cb) < 1e-12 and cases.rel_err(ka, kb) < 1e-12
    assert torch.equal(a._sgn, b._sgn)
    assert cases.rel_err(cnp(a._c2), cnp(b._c2)) < 1e-10 and cases.rel_err(cnp(a.y), cnp(b.y)) < 1e-13


def test_whole_loop_in_normal_mode_coordinates_equals_the_product_with_phi():
    """sc_hk_run_modal (round 4): run() of a constant dense Hessian with the monodromy blocks in normal-mode coordinates (per-mode
    2 x 2 step matrices, transformed prefactor constants, two changes of basis around the loop) against the same loop with the
    product with Phi(dt): reference goldens through both, state and correlation functions to 1e-12, signs exact"""
    from tests.engine_cases import engine_potential, engine_propagator
    g = cases.load("hk_methylium")
    pot = engine_potential(g)
    nt, dt, E0 = int(g["nt"]), float(g["dt"]), float(g["E0"])
    out = []
    for frm in (16, 10 ** 9):
        prop = engine_propagator(g)
        prop.normal_modes_from = frm
        c, k = prop.run(pot, dt, nt, E0)
        prop.synchronize()
        assert bool(prop.__dict__.get("_modal_cache")) == (frm == 16)
        assert cases.rel_err(c, g["cauto"]) < TOL and cases.rel_err(k, g["kic"]) < TOL
        assert cases.rel_err(cnp(prop.y), g[f"y_{nt}"]) < TOL
        out.append((c, k, cnp(prop.y), cnp(prop._c2), cnp(prop._sgn)))
    a, b = out
    assert cases.rel_err(a[0], b[0]) < 1e-12 and cases.rel_err(a[1], b[1]) < 1e-12
    assert cases.rel_err(a[2], b[2]) < 1e-12 and cases.rel_err(a[3], b[3]) < 1e-11
    assert np.array_equal(a[4], b[4])
    # fewer steps than normal_modes_from: the changes of basis would cost more than they save
    short = engine_propagator(g)
    short.run(pot, dt, 8, E0)
    assert not short.__dict__.get("_modal_cache")
