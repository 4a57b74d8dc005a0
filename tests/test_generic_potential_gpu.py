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
