"""Generic Python potentials ("generic Python potentials take the unfused path"): an object that
only implements the reference's potential protocol in torch is evaluated by its own code at the RK4 stage points, the
rest of the step runs in HIP.  Checked against the CPU oracle driving the very same class."""
import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu
torch.set_default_dtype(torch.float64)


class CoupledQuarticPotential(object):
    """V = 1/2 sum_a w_a^2 r_a^2 + lam (sum_a r_a r_{a+1})^2 : non-separable, Hessian dense and position dependent"""

    def __init__(self, omega, lam, masses, nac):
        self.omega, self.lam, self._masses, self.nac = omega, lam, masses, nac

    def dimensions(self):
        return len(self.omega)

    def masses(self):
        return self._masses

    def harmonic_approximation(self, r):
        w2 = (self.omega ** 2).to(r.device).unsqueeze(1)
        d, n = r.shape
        s = torch.sum(r[:-1] * r[1:], dim=0)                                  # (n,)
        ds = torch.zeros_like(r)                                              # d s / d r_a
        ds[:-1] += r[1:]
        ds[1:] += r[:-1]
        V = 0.5 * torch.sum(w2 * r * r, dim=0) + self.lam * s * s
        grad = w2 * r + 2.0 * self.lam * s * ds
        hess = torch.zeros((d, d, n), dtype=r.dtype, device=r.device)
        idx = torch.arange(d, device=r.device)
        hess[idx, idx] = w2.expand(-1, n).clone()
        hess += 2.0 * self.lam * ds.unsqueeze(1) * ds.unsqueeze(0)
        off = 2.0 * self.lam * s
        hess[idx[:-1], idx[1:]] += off
        hess[idx[1:], idx[:-1]] += off
        return V, grad, hess

    def derivative_coupling_1st(self, r):
        return self.nac.to(r.device).unsqueeze(1).expand(-1, r.shape[1])

    def derivative_coupling_2nd(self, r):
        return torch.zeros_like(r)


@pytest.mark.parametrize("kind", ["HK", "WM"])
def test_generic_python_potential_matches_oracle(kind):
    from oracle import sc_oracle as orc
    from semiclassical_amd import propagators as PR
    rng = np.random.default_rng(21)
    D, n, nt, dt = 5, 160, 15, 1.5
    omega = torch.from_numpy(np.sort(rng.uniform(700, 2600, D)) / 219474.63)
    masses = torch.from_numpy(rng.uniform(0.8, 1.6, D))
    nac = torch.from_numpy(rng.normal(0, 1e-3, D))
    pot = CoupledQuarticPotential(omega, 2.0e-6, masses, nac)
    q0 = torch.from_numpy(rng.uniform(-6.0, 6.0, D))
    p0 = 0.0 * q0
    G = torch.diag(omega * masses)
    E0 = float(0.5 * omega.sum())
    if kind == "HK":
        ref, prop = orc.HKOracle(G, G), PR.HermanKlukPropagator(G, G, device="cuda")
    else:
        ref, prop = orc.WMOracle(G, G, 80.0, 80.0), PR.WaltonManolopoulosPropagator(G, G, 80.0, 80.0, device="cuda")
    torch.manual_seed(9)
    ref.initial_conditions(q0, p0, G, ntraj=n)
    rc, rk = orc.run_loop(ref, pot, dt, nt, E0)
    prop.set_initial_conditions(q0, p0, G, ref.zi, ref.probi)
    c, k = prop.run(pot, dt, nt, E0)
    assert np.abs(rc[-1] - rc[0]) > 1e-3                     # the dynamics is not trivial
    assert cases.rel_err(c, rc) < 1e-9 and cases.rel_err(k, rk) < 1e-9
    y_ref = ref.y.numpy()
    assert cases.rel_err(prop.y.cpu().numpy(), y_ref) < 1e-9


@pytest.mark.parametrize("D,dense_gamma", [(70, False), (70, True), (100, False), (100, True), (130, True)])
def test_more_than_64_dimensions_through_the_dense_path(D, dense_gamma):
    """Beyond the fused kernels (D > 64) every potential takes the dense path (its own torch code at the stage points):
    MFMA monodromy kernel with the RK4 sums in a global scratch and a panelised prefactor up to D = 96, the
    any-dimension kernels (global scratch, pivoted LU on global memory) beyond -- the reference has no size limit
    (propagators.py:313-383).  Anharmonic AS (Morse) model, diagonal or rotated width matrices, against the CPU oracle."""
    from oracle import sc_oracle as orc
    from semiclassical_amd import potentials as P, propagators as PR
    rng = np.random.default_rng(D)
    n, nt, dt = (24, 4, 2.0) if D <= 96 else (10, 3, 2.0)
    omega = torch.from_numpy(np.sort(rng.uniform(400, 3200, D)) / 219474.63)
    S = torch.from_numpy(rng.uniform(0.01, 0.1, D) * rng.choice([-1, 1], D))
    nac = torch.from_numpy(rng.normal(0, 1e-4, D))
    chi = torch.full((D,), 0.02)
    q0 = torch.sqrt(2 * abs(S) / omega) * torch.sign(S)
    p0 = 0.0 * q0
    if dense_gamma:
        Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
        G = torch.from_numpy(Q @ np.diag(omega.numpy() * rng.uniform(0.8, 1.25, D)) @ Q.T)
        G = 0.5 * (G + G.T)
    else:
        G = torch.diag(omega)
    E0 = float(0.5 * omega.sum())
    ref = orc.HKOracle(G, G)
    torch.manual_seed(2)
    ref.initial_conditions(q0, p0, G, ntraj=n)
    rc, rk = orc.run_loop(ref, orc.MorseOracle(omega, chi.clone(), nac), dt, nt, E0)
    prop = PR.HermanKlukPropagator(G, G, device="cuda")
    prop.set_initial_conditions(q0, p0, G, ref.zi, ref.probi)
    c, k = prop.run(P.MorsePotential(omega, chi.clone(), nac), dt, nt, E0)
    assert cases.rel_err(prop.y.cpu().numpy(), ref.y.numpy()) < 1e-9
    assert cases.rel_err(c, rc) < 1e-8 and cases.rel_err(k, rk) < 1e-8


# ---------------------------------------------------------------------------------------------------------------------
# position-dependent derivative couplings (reference propagators.py:880-903, 1685-1714: the couplings are evaluated at
# the initial AND the current points of every trajectory; the reference's own potentials return constants)
# ---------------------------------------------------------------------------------------------------------------------

def _varying_tau1(nac, r):
    """tau1_a(r) = nac_a (1 + 0.4 r_a + 0.15 r_{a+1} + 0.05 r_a^2): changes along every trajectory"""
    nac = nac.to(r.device).unsqueeze(1)
    return nac * (1.0 + 0.4 * r + 0.15 * torch.roll(r, -1, dims=0) + 0.05 * r * r)


def _varying_tau2(nac, r):
    nac = nac.to(r.device).unsqueeze(1)
    return nac * (0.4 + 0.1 * r) * 1e-2


class QuarticWithVaryingCoupling(CoupledQuarticPotential):
    def derivative_coupling_1st(self, r):
        return _varying_tau1(self.nac, r)

    def derivative_coupling_2nd(self, r):
        return _varying_tau2(self.nac, r)


@pytest.mark.parametrize("kind", ["HK", "WM"])
@pytest.mark.parametrize("surface", ["generic", "morse", "harmonic-dense"])
def test_position_dependent_couplings_match_oracle(kind, surface):
    """k_ic(t) with couplings that depend on the position, against the oracle driving the same coupling functions: a generic
    Python surface (unfused step), the Morse surface (fused separable step kernel, whole-loop / graph paths must stand
    back) and the constant-Hessian molecular surface (register kernels; WM must leave wm_small_kernel for the kernel that
    takes per-trajectory coupling vectors)."""
    from oracle import sc_oracle as orc
    from semiclassical_amd import potentials as P, propagators as PR
    rng = np.random.default_rng(33)
    D, n, nt, dt = 5, 150, 12, 1.5
    omega = torch.from_numpy(np.sort(rng.uniform(700, 2600, D)) / 219474.63)
    nac = torch.from_numpy(rng.normal(0, 1e-3, D))
    if surface == "generic":
        masses = torch.from_numpy(rng.uniform(0.8, 1.6, D))
        pot = ref_pot = QuarticWithVaryingCoupling(omega, 2.0e-6, masses, nac)
        G = torch.diag(omega * masses)
        q0 = torch.from_numpy(rng.uniform(-6.0, 6.0, D))
    elif surface == "morse":
        chi = torch.full((D,), 0.02)

        class Eng(P.MorsePotential):
            def derivative_coupling_1st(self, r):
                return _varying_tau1(nac, r)

            def derivative_coupling_2nd(self, r):
                return _varying_tau2(nac, r)

        class Orc(orc.MorseOracle):
            def derivative_coupling_1st(self, r):
                return _varying_tau1(nac, r)

            def derivative_coupling_2nd(self, r):
                return _varying_tau2(nac, r)
        pot, ref_pot = Eng(omega, chi.clone(), nac), Orc(omega, chi.clone(), nac)
        G = torch.diag(omega)
        q0 = torch.from_numpy(rng.uniform(-6.0, 6.0, D))
    else:
        g = cases.load("wm_methylium" if kind == "WM" else "hk_methylium")
        D, dt = 12, float(g["dt"])
        nac = cases.T(g["nac0"])
        args = (g["pos0"], g["energy0"], g["grad0"], g["hess0"], g["masses"], g["nac0"])

        class Eng(P.MolecularHarmonicPotential):
            def derivative_coupling_1st(self, r):
                return _varying_tau1(nac, r)

            def derivative_coupling_2nd(self, r):
                return _varying_tau2(nac, r)

        class Orc(orc.MolecularHarmonicOracle):
            def derivative_coupling_1st(self, r):
                return _varying_tau1(nac, r)

            def derivative_coupling_2nd(self, r):
                return _varying_tau2(nac, r)
        pot = Eng.from_arrays(*args, origin=float(g["origin"]))
        ref_pot = Orc(*args, origin=float(g["origin"]))
        G = cases.T(g["Gamma_i"])
        q0 = cases.T(g["q0"])
        n = 96
    p0 = 0.0 * q0
    E0 = 0.01
    G0 = cases.T(g["Gamma_0"]) if surface == "harmonic-dense" else G
    if kind == "HK":
        ref, prop = orc.HKOracle(G, G), PR.HermanKlukPropagator(G, G, device="cuda")
    else:
        ab = 1.0e4 if surface == "harmonic-dense" else 80.0
        ref, prop = orc.WMOracle(G, G, ab, ab), PR.WaltonManolopoulosPropagator(G, G, ab, ab, device="cuda")
    torch.manual_seed(4)
    ref.initial_conditions(q0, p0, G0, ntraj=n)
    rc, rk = orc.run_loop(ref, ref_pot, dt, nt, E0)
    prop.set_initial_conditions(q0, p0, G0, ref.zi, ref.probi)
    c, k = prop.run(pot, dt, nt, E0)
    assert cases.rel_err(c, rc) < 1e-9 and cases.rel_err(k, rk) < 1e-8, (cases.rel_err(c, rc), cases.rel_err(k, rk))
    # ... and the step-at-a-time API gives the same numbers as run()
    prop.set_initial_conditions(q0, p0, G0, ref.zi, ref.probi)
    for t in range(3):
        assert abs(prop.autocorrelation(E0) - rc[t]) < 1e-9 * abs(rc[t]) + 1e-14
        assert abs(prop.ic_correlation(pot, E0) - rk[t]) < 1e-8 * np.abs(rk).max()
        prop.step(pot, dt)
