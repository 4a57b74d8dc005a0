"""CPU oracle for the Herman-Kluk / Walton-Manolopoulos propagator loop.

TEST INFRASTRUCTURE ONLY.  This module is a torch-CPU restatement of the
algorithm in the reference's ``semiclassical/propagators.py`` and
``semiclassical/potentials.py``.  It exists to *check* the HIP engine in
``semiclassical_amd`` and to serve as the timed CPU baseline in ``bench.py``;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  The product path never does.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference
itself (in the build container, where ``/root/reference`` exists) and stores
its inputs/outputs as ``tests/golden/*.npz``; ``tests/test_oracle.py`` checks
this restatement against those vectors to <= 1e-11.

The restatement deliberately keeps the reference's *eager op sequence* --
``(rows, n)`` state with the trajectory index fastest, a dense ``(D, D, n)``
Hessian per RK4 stage, ``einsum`` contractions, ``cat`` of the slopes, a
batched complex ``det`` -- so that its wall time is a fair stand-in for the
reference's CPU path.  All citations are relative to
``/root/reference``.
"""

import numpy as np
import torch

hbar = 1.0            # semiclassical/units.py:8
ZERO = 1.0e-8         # semiclassical/propagators.py:16

C128 = torch.complex128


def eigh(A):
    """torch.symeig of the reference == torch.linalg.eigh (upper triangle)."""
    return torch.linalg.eigh(A, UPLO='U')


def sym_sqrtm(A):
    """A^{1/2} and pseudo-inverse A^{-1/2} as complex (D,D); propagators.py:25-59."""
    w, V = eigh(A)
    keep = abs(w) > ZERO
    wc, Vc = w.type(C128), V.type(C128)
    root = torch.einsum('ij,j,kj->ik', Vc, torch.sqrt(wc), Vc)
    iroot = torch.einsum('ij,j,kj->ik', Vc[:, keep], 1.0 / torch.sqrt(wc[keep]), Vc[:, keep])
    return root, iroot


def is_symmetric_non_negative(A, eps=1.0e-6):
    """propagators.py:61-82"""
    if torch.sum(abs(A - A.T)) / torch.sum(abs(A)) > eps:
        return False
    w, _ = eigh(A)
    return bool((w >= -ZERO).all())


# --------------------------------------------------------------------------
# potentials (potentials.py)
# --------------------------------------------------------------------------

class _Separable(object):
    """shared plumbing of the 1-D separable model potentials"""

    def dimensions(self):
        return self._dim

    def masses(self):
        return torch.ones(self._dim)

    def harmonic_approximation(self, r):
        return self._energy(r), self._gradient(r), self._hessian(r)

    def _dense_diag(self, d):
        """(D,n) diagonal -> dense (D,D,n) Hessian, potentials.py:320-325"""
        dim, n = d.shape
        hess = torch.zeros((dim, dim, n), dtype=d.dtype)
        torch.diagonal(hess, dim1=0, dim2=1)[...] = d.transpose(0, 1)
        return hess

    def derivative_coupling_2nd(self, r):
        return torch.zeros_like(r)


class MorseOracle(_Separable):
    """anharmonic AS ground-state potential, potentials.py:208-397"""

    def __init__(self, omega, chi, nac):
        omega, chi, nac = (torch.as_tensor(x, dtype=torch.float64).clone() for x in (omega, chi, nac))
        self.omega, self.nac = omega, nac
        self.harmonic = bool((chi == 0.0).all())
        if not self.harmonic:
            chi[chi == 0.0] += 1.0e-4         # potentials.py:250
        self.chi = chi
        self.a = torch.sqrt(2 * omega * chi)  # potentials.py:254
        self.D = 0.25 * omega / chi           # potentials.py:255
        self._dim = omega.shape[0]

    def _col(self, v, r):
        return v.unsqueeze(1).expand_as(r)

    def _energy(self, r):
        if self.harmonic:
            return torch.sum(0.5 * self._col(self.omega, r) ** 2 * r ** 2, 0)
        a, D = self._col(self.a, r), self._col(self.D, r)
        return torch.sum(D * (1.0 - torch.exp(-a * r)) ** 2, 0)

    def _gradient(self, r):
        if self.harmonic:
            return self._col(self.omega, r) ** 2 * r
        a, D = self._col(self.a, r), self._col(self.D, r)
        return 2 * a * D * torch.exp(-a * r) * (1.0 - torch.exp(-a * r))

    def _hessian(self, r):
        if self.harmonic:
            return self._dense_diag(self._col(self.omega, r) ** 2)
        a, D = self._col(self.a, r), self._col(self.D, r)
        return self._dense_diag(2 * a ** 2 * D * torch.exp(-a * r) * (2 * torch.exp(-a * r) - 1.0))

    def derivative_coupling_1st(self, r):
        return self._col(self.nac, r)


class NonHarmonicOracle(_Separable):
    """eps*Morse + (1-eps)*harmonic of Herman & Kluk 1986, potentials.py:25-204"""

    def __init__(self, eps=(0.975,), b=(12.0 ** (-0.5),)):
        self.eps = torch.as_tensor(eps, dtype=torch.float64)
        self.b = torch.as_tensor(b, dtype=torch.float64)
        self._dim = self.eps.shape[0]

    def _row(self, v, r):
        # the reference broadcasts with unsqueeze(0) (potentials.py:77), which is
        # only meaningful for dim == 1; kept as is.
        return v.unsqueeze(0).expand_as(r)

    def _energy(self, r):
        eps, b = self._row(self.eps, r), self._row(self.b, r)
        return torch.sum(eps / (2 * b ** 2) * (1.0 - torch.exp(-b * r)) ** 2 + (1 - eps) * 0.5 * r ** 2, 0)

    def _gradient(self, r):
        eps, b = self._row(self.eps, r), self._row(self.b, r)
        return eps / b * (torch.exp(-b * r) - torch.exp(-2 * b * r)) + (1 - eps) * r

    def _hessian(self, r):
        eps, b = self._row(self.eps, r), self._row(self.b, r)
        return self._dense_diag(eps * (2 * torch.exp(-2 * b * r) - torch.exp(-b * r)) + (1 - eps))

    def derivative_coupling_1st(self, r):
        return torch.ones_like(r)


class MolecularHarmonicOracle(object):
    """second-order expansion around pos0, potentials.py:529-638 (arrays instead of fchk objects)"""

    def __init__(self, pos0, energy0, grad0, hess0, masses, nac0, origin=0.0):
        t = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float64)
        self.pos0, self.grad0, self.hess0 = t(pos0), t(grad0), t(hess0)
        self.energy0 = t(energy0)
        self._masses, self.nac0 = t(masses), t(nac0)
        self._dim = self._masses.shape[0]
        self._origin = float(origin)

    def dimensions(self):
        return self._dim

    def masses(self):
        return self._masses

    def harmonic_approximation(self, r):
        dim, n = r.shape
        dr = r - self.pos0.unsqueeze(1).expand_as(r)
        vpot = (self.energy0
                + torch.einsum('in,i->n', dr, self.grad0)
                + 0.5 * torch.einsum('in,ij,jn->n', dr, self.hess0, dr))
        grad = self.grad0.unsqueeze(1).expand_as(r) + torch.einsum('ij,jn->in', self.hess0, dr)
        hess = self.hess0.unsqueeze(2).expand(-1, -1, n)
        return vpot - self._origin, grad, hess

    def derivative_coupling_1st(self, r):
        return self.nac0.unsqueeze(1).expand_as(r)

    def derivative_coupling_2nd(self, r):
        return torch.zeros_like(r)


# --------------------------------------------------------------------------
# equations of motion + RK4 (propagators.py:86-119, 296-398)
# --------------------------------------------------------------------------

class EomOracle(object):
    def __init__(self):
        self.history = []
        self.en_mean = None

    def f(self, t, y, potential):
        d = potential.dimensions()
        m = potential.masses()
        q, p, Mqq, Mqp, Mpq, Mpp, _ = torch.split(y, [d, d, d * d, d * d, d * d, d * d, 1])
        Mqq, Mqp, Mpq, Mpp = (X.view(d, d, -1) for X in (Mqq, Mqp, Mpq, Mpp))
        vpot, grad, hess = potential.harmonic_approximation(q)
        m3 = m.unsqueeze(1).unsqueeze(2)
        dMqq = Mpq / m3.expand_as(Mpq)
        dMpq = -torch.einsum('ag...,gb...->ab...', hess, Mqq)
        dMqp = Mpp / m3.expand_as(Mpp)
        dMpp = -torch.einsum('ag...,gb...->ab...', hess, Mqp)
        dq = p / m.unsqueeze(1).expand_as(p)
        dp = -grad
        tkin = 0.5 * torch.sum(p ** 2 / m.unsqueeze(1).expand_as(p), 0)
        dS = tkin - vpot
        self.en_mean = torch.mean(tkin + vpot)       # Q2: value of the *last evaluated stage*
        return torch.cat((dq, dp,
                          dMqq.reshape(d * d, -1), dMqp.reshape(d * d, -1),
                          dMpq.reshape(d * d, -1), dMpp.reshape(d * d, -1),
                          dS.reshape(1, -1)), 0)

    def check_energy_conservation(self, tol=1.0e-2):
        self.history.append(self.en_mean)
        if len(self.history) > 1:
            change = abs(self.history[1] - self.history[0])
            if change > tol:
                raise RuntimeError("average energy of classical trajectories is not conserved, "
                                   f"change= {change} Hartree")
            self.history.pop(0)


def rk4_step(eom, y, t, h, potential):
    k1 = eom.f(t, y, potential)
    k2 = eom.f(t + 0.5 * h, y + 0.5 * h * k1, potential)
    k3 = eom.f(t + 0.5 * h, y + 0.5 * h * k2, potential)
    k4 = eom.f(t + h, y + h * k3, potential)
    return y + h / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)


# --------------------------------------------------------------------------
# coherent-state overlaps (propagators.py:124-240)
# --------------------------------------------------------------------------

class OverlapOracle(object):
    def __init__(self, Gi, Gj):
        self.dim = Gi.shape[0]
        ei, _ = eigh(Gi)
        ej, _ = eigh(Gj)
        self.rank = int(torch.count_nonzero(abs(ei) > ZERO))
        assert self.rank == int(torch.count_nonzero(abs(ej) > ZERO))
        self.detGi = torch.prod(ei[abs(ei) > ZERO])
        self.detGj = torch.prod(ej[abs(ej) > ZERO])
        eij, Vij = eigh(Gi + Gj)
        keep = abs(eij) > ZERO
        self.iGij = torch.einsum('ij,j,kj->ik', Vij[:, keep], 1.0 / eij[keep], Vij[:, keep])
        self.detGij = torch.prod(eij[keep])
        self.Gi_iGij_Gj = Gi @ self.iGij @ Gj
        self.Gj_iGij = Gj @ self.iGij
        self.fac = torch.sqrt(2.0 ** self.rank * torch.sqrt(self.detGi) * torch.sqrt(self.detGj) / self.detGij)

    def __call__(self, qi, pi, qj, pj):
        """<qi,pi,Gi|qj,pj,Gj> for batches (D,ni) x (D,nj) -> (ni,nj)"""
        if qi.dim() == 1:
            qi, pi = qi.unsqueeze(1), pi.unsqueeze(1)
        if qj.dim() == 1:
            qj, pj = qj.unsqueeze(1), pj.unsqueeze(1)
        ni, nj = qi.shape[1], qj.shape[1]
        qi, pi = qi.unsqueeze(2).expand(-1, -1, nj), pi.unsqueeze(2).expand(-1, -1, nj)
        qj, pj = qj.unsqueeze(1).expand(-1, ni, -1), pj.unsqueeze(1).expand(-1, ni, -1)
        dq, dp = qj - qi, pj - pi
        return self.fac * torch.exp(
            -0.5 * torch.einsum('aij,ab,bij->ij', dq, self.Gi_iGij_Gj, dq)
            - 0.5 / hbar ** 2 * torch.einsum('aij,ab,bij->ij', dp, self.iGij, dp)
            - 1j / hbar * torch.einsum('aij,aij->ij', pj, dq)
            + 1j / hbar * torch.einsum('aij,ab,bij->ij', dq, self.Gj_iGij, dp))


# --------------------------------------------------------------------------
# Herman-Kluk propagator (propagators.py:407-1066)
# --------------------------------------------------------------------------

class SignTracker(object):
    """branch tracking of sqrt(z(t)), propagators.py:1006-1066"""

    def __init__(self):
        self.state = {}

    def track(self, key, z):
        if key not in self.state:
            self.state[key] = {"signs": torch.ones_like(z), "previous": z}
        st = self.state[key]
        z1, z2 = st["previous"], z
        flip = (z1.real < 0) & (z2.real < 0) & (z1.imag * z2.imag < 0)
        st["signs"][flip] *= -1.0
        st["previous"] = z

    def signs(self, key):
        return self.state[key]["signs"]


def ic_sampling_matrices(Gamma_i, Gamma_0):
    """the small host-side linear algebra of initial_conditions, propagators.py:493-531

    Returns U (D,d') complex, iGi0 (D,D), iLz (2d',2D), detLz, d'
    """
    Gi0 = Gamma_0 + Gamma_i
    wp, Vp = eigh(Gi0)
    nzp = wp > ZERO
    U = Vp[:, nzp].type(C128)
    iGi0 = torch.einsum('ij,j,kj->ik', Vp[:, nzp], 1.0 / wp[nzp], Vp[:, nzp])
    iLp = torch.einsum('i,ji->ij', torch.sqrt(wp[nzp] / 2), Vp[:, nzp])
    wq, Vq = eigh(Gamma_i @ iGi0 @ Gamma_0)
    nzq = wq > ZERO
    iLq = torch.einsum('i,ji->ij', 1.0 / torch.sqrt(2 * wq[nzq]), Vq[:, nzq])
    dprime = int(torch.count_nonzero(nzp))
    assert dprime == int(torch.count_nonzero(nzq))
    iLz = torch.block_diag(iLq, iLp)
    detLz = torch.prod(2 * torch.sqrt(wq[nzq] / wp[nzp]))
    return U, iGi0, iLz, detLz, dprime


class HKOracle(object):
    def __init__(self, Gamma_i, Gamma_t):
        assert is_symmetric_non_negative(Gamma_i) and is_symmetric_non_negative(Gamma_t)
        self.Gamma_i, self.Gamma_t = Gamma_i, Gamma_t
        self.sqGi, self.isqGi = sym_sqrtm(Gamma_i)
        self.sqGt, self.isqGt = sym_sqrtm(Gamma_t)
        self.eom = EomOracle()
        self.tracker = SignTracker()

    # -- initial conditions ------------------------------------------------
    def initial_conditions(self, q0, p0, Gamma_0, ntraj=5000, xi=None):
        """sample (or, with ``xi`` given, reproduce) the initial phase-space points.

        ``xi`` (2d', n) are the standard-normal deviates of propagators.py:537-539;
        passing the reference's own draw pins zi/probi exactly.
        """
        assert Gamma_0.shape == self.Gamma_i.shape
        assert is_symmetric_non_negative(Gamma_0)
        d, n = q0.shape[0], ntraj
        self.U, self.iGi0, iLz, detLz, dprime = ic_sampling_matrices(self.Gamma_i, Gamma_0)
        if xi is None:
            xi = torch.distributions.Normal(torch.zeros(2 * dprime), torch.ones(2 * dprime)).sample((n,)).T
        z0 = torch.cat((q0, p0))
        zi = z0.unsqueeze(1) + torch.einsum('ji,jn->in', iLz, xi)
        probi = detLz / (2 * np.pi) ** d * torch.exp(-0.5 * torch.einsum('in,in->n', xi, xi))
        self.set_initial_conditions(q0, p0, Gamma_0, zi, probi)

    def set_initial_conditions(self, q0, p0, Gamma_0, zi, probi):
        """start from given phase-space points zi (2D,n) with sampling densities probi (n,)"""
        d, n = q0.shape[0], zi.shape[1]
        if not hasattr(self, "U"):
            self.U, self.iGi0, _, _, _ = ic_sampling_matrices(self.Gamma_i, Gamma_0)
        y = torch.zeros((2 * d + 4 * d * d + 1, n))
        z, Mqq, _, _, Mpp, _ = torch.split(y, [2 * d, d * d, d * d, d * d, d * d, 1])
        eye = torch.eye(d).unsqueeze(2).expand(-1, -1, n)
        Mqq.view(d, d, -1)[...] = eye
        Mpp.view(d, d, -1)[...] = eye
        z[...] = zi
        self.dim, self.ntraj = d, n
        self.q0, self.p0, self.Gamma_0 = q0, p0, Gamma_0
        self.zi, self.probi, self.y = zi, probi, y
        self.c = torch.ones(n, dtype=C128)
        self._prepare()
        self.t = 0.0
        self._prefactor()

    def _prepare(self):
        self.csoi0 = OverlapOracle(self.Gamma_i, self.Gamma_0)
        self.csot0 = OverlapOracle(self.Gamma_t, self.Gamma_0)
        self.csott = OverlapOracle(self.Gamma_t, self.Gamma_t)

    # -- accessors -----------------------------------------------------------
    def _blocks(self):
        d = self.dim
        return torch.split(self.y, [d, d, d * d, d * d, d * d, d * d, 1])

    def initial_positions_and_momenta(self):
        return torch.split(self.zi, [self.dim, self.dim])

    def current_positions_and_momenta(self):
        return self._blocks()[:2]

    def classical_action(self):
        return self._blocks()[-1].squeeze()

    def monodromy_matrices(self):
        d = self.dim
        return tuple(X.view(d, d, -1) for X in self._blocks()[2:6])

    def semiclassical_prefactor(self):
        return self.tracker.signs("prefactorC") * self.c

    # -- time step -------------------------------------------------------------
    def step(self, potential, dt):
        assert self.dim == potential.dimensions()
        self.y = rk4_step(self.eom, self.y, self.t, dt, potential)
        self.eom.check_energy_conservation()
        self._prefactor()
        self.t += dt

    def _prefactor(self):
        """HK prefactor, eqn (29); propagators.py:951-1004"""
        Mqq, Mqp, Mpq, Mpp = (X.type(C128) for X in self.monodromy_matrices())
        mat = 0.5 * (torch.einsum('ai,ijn,jb->abn', self.sqGt, Mqq, self.isqGi)
                     + torch.einsum('ai,ijn,jb->abn', self.isqGt, Mpp, self.sqGi)
                     - 1j * hbar * torch.einsum('ai,ijn,jb->abn', self.sqGt, Mqp, self.sqGi)
                     + 1j / hbar * torch.einsum('ai,ijn,jb->abn', self.isqGt, Mpq, self.isqGi))
        mat = torch.einsum('ia,ijn,jb->abn', self.U, mat, self.U)
        c2 = torch.det(mat.permute(2, 0, 1))
        self.c2 = c2
        self.c = torch.sqrt(c2)
        self.tracker.track("prefactorC", c2)

    # -- correlation functions ---------------------------------------------------
    def autocorrelation_qp(self):
        qi, pi = self.initial_positions_and_momenta()
        vi = self.csoi0(qi, pi, self.q0, self.p0).squeeze()
        qt, pt = self.current_positions_and_momenta()
        vt = self.csot0(qt, pt, self.q0, self.p0).squeeze()
        return vt.conj() * vi * self.semiclassical_prefactor() * torch.exp(1j / hbar * self.classical_action())

    def _mc_weight(self):
        return self.ntraj * self.probi * (2 * np.pi * hbar) ** self.dim

    def autocorrelation(self, energy0_es=0.0):
        cauto = torch.sum(self.autocorrelation_qp() / self._mc_weight())
        return (cauto * torch.exp(1j / hbar * self.t * torch.tensor(energy0_es))).item()

    def ic_correlation(self, potential, energy0_es=0.0):
        cauto_qp = self.autocorrelation_qp()
        q, p = self.initial_positions_and_momenta()
        Q, P = self.current_positions_and_momenta()
        q0 = self.q0.unsqueeze(1).expand_as(q)
        p0 = self.p0.unsqueeze(1).expand_as(p)
        im = 1.0 / potential.masses()
        n1q = -hbar ** 2 * torch.einsum('k,kn->kn', im, potential.derivative_coupling_1st(q))
        n1Q = -hbar ** 2 * torch.einsum('k,kn->kn', im, potential.derivative_coupling_1st(Q))
        n2q = -hbar ** 2 * 0.5 * torch.einsum('k,kn->n', im, potential.derivative_coupling_2nd(q))
        n2Q = -hbar ** 2 * 0.5 * torch.einsum('k,kn->n', im, potential.derivative_coupling_2nd(Q))
        PI = p0 + torch.einsum('ij,jk,kn->in', self.Gamma_0, self.iGi0, P - p0)
        pi = p0 + torch.einsum('ij,jk,kn->in', self.Gamma_0, self.iGi0, p - p0)
        R = torch.einsum('ij,jk,kl->il', self.Gamma_0, self.iGi0, self.Gamma_i)
        nacQ = n2Q + (torch.einsum('in,ij,jn->n', q0 - Q, R, n1Q) - 1j / hbar * torch.einsum('in,in->n', PI, n1Q))
        nacq = n2q + (torch.einsum('in,ij,jn->n', q0 - q, R, n1q) + 1j / hbar * torch.einsum('in,in->n', pi, n1q))
        kic = (1.0 / hbar ** 2 * torch.exp(1j / hbar * self.t * torch.tensor(energy0_es))
               * nacQ * nacq * cauto_qp)
        return torch.sum(kic / self._mc_weight()).item()


# --------------------------------------------------------------------------
# Walton-Manolopoulos propagator (propagators.py:1077-1719)
# --------------------------------------------------------------------------

class WMOracle(HKOracle):
    def __init__(self, Gamma_i, Gamma_t, alpha, beta):
        super().__init__(Gamma_i, Gamma_t)
        self.alpha, self.beta = torch.tensor(float(alpha)), torch.tensor(float(beta))

    def _prepare(self):
        pdet = lambda G, s: torch.prod((lambda e: e[abs(e) > ZERO] / s)(eigh(G)[0]))
        self.detG0 = pdet(self.Gamma_0, np.pi)
        self.detGi = pdet(self.Gamma_i, np.pi)
        self.detGt = pdet(self.Gamma_t, np.pi)
        self.detGi0 = pdet(self.Gamma_0 + self.Gamma_i, 2 * np.pi)
        e0, V0 = eigh(self.Gamma_0)
        keep = e0 > ZERO
        self.iGamma_0 = torch.einsum('ij,j,kj->ik', V0[:, keep], 1.0 / e0[keep], V0[:, keep])

    def _expand_L(self):
        """gradient and (truncated) Hessian of i/hbar*S w.r.t. z=(q,p); propagators.py:1132-1193"""
        Mqq, Mqp, Mpq, Mpp = self.monodromy_matrices()
        q, p = self.initial_positions_and_momenta()
        Q, P = self.current_positions_and_momenta()
        dSdq = torch.einsum('ijn,in->jn', Mqq, P) - p
        dSdp = torch.einsum('ijn,in->jn', Mqp, P)
        gradL = 1j / hbar * torch.cat((dSdq, dSdp), dim=0)
        Sqq = torch.einsum('ijn,ikn->jkn', Mpq, Mqq)
        Sqp = torch.einsum('ijn,ikn->jkn', Mpq, Mqp)
        Spq = torch.einsum('ijn,ikn->jkn', Mqp, Mpq)
        Spp = torch.einsum('ijn,ikn->jkn', Mqp, Mpp)
        hessL = 1j / hbar * torch.cat((torch.cat((Sqq, Sqp), dim=1), torch.cat((Spq, Spp), dim=1)), dim=0)
        return gradL, hessL

    def _prefactor(self):
        super()._prefactor()
        d, n = self.dim, self.ntraj
        Mqq, Mqp, Mpq, Mpp = self.monodromy_matrices()
        q, p = self.initial_positions_and_momenta()
        Q, P = self.current_positions_and_momenta()
        gradL, hessL = self._expand_L()
        Mqz, Mpz = torch.cat((Mqq, Mqp), dim=1), torch.cat((Mpq, Mpp), dim=1)
        Eqz = torch.cat((torch.eye(d), torch.zeros(d, d)), dim=1).unsqueeze(2).expand_as(Mqz)
        Epz = torch.cat((torch.zeros(d, d), torch.eye(d)), dim=1).unsqueeze(2).expand_as(Mpz)
        filinov = torch.block_diag(self.alpha * self.Gamma_0, self.beta * self.iGamma_0).unsqueeze(2).expand(-1, -1, n)
        # eqn (50)
        A = 2 * filinov - hessL + (
            torch.einsum('jin,jk,kln->iln', Mqz, self.Gamma_t, Mqz)
            + torch.einsum('jin,jk,kln->iln', Eqz, self.Gamma_i, Eqz)
            + 2j / hbar * (torch.einsum('jin,jkn->ikn', Mpz, Mqz) - torch.einsum('jin,jkn->ikn', Epz, Eqz)))
        U2 = torch.block_diag(self.U, self.U)
        A = torch.einsum('ia,ijn,jb->abn', U2, A, U2)
        iA = torch.inverse(A.permute(2, 0, 1)).permute(1, 2, 0)
        iA = torch.einsum('ai,ijn,bj->abn', U2, iA, U2)
        BQ = torch.einsum('ij,jkn->ikn', self.Gamma_t, Mqz) + 1j / hbar * Mpz          # (53)
        Bq = torch.einsum('ij,jkn->ikn', self.Gamma_i, Eqz) - 1j / hbar * Epz          # (54)
        b0 = gradL - 1j / hbar * (torch.einsum('jin,jn->in', Mqz, P) - torch.einsum('jin,jn->in', Eqz, p))  # (55)
        Gt = self.Gamma_t.unsqueeze(2).expand(-1, -1, n) - torch.einsum('ijn,jkn,lkn->iln', BQ, iA, BQ)      # (57)
        Gti = torch.einsum('ijn,jkn,lkn->iln', BQ, iA, Bq)                                                     # (59)
        pi_t = P - 1j * hbar * torch.einsum('ijn,jkn,kn->in', BQ, iA, b0)                                      # (60)
        pi_i = p + 1j * hbar * torch.einsum('ijn,jkn,kn->in', Bq, iA, b0)
        q0 = self.q0.unsqueeze(1).expand_as(q)
        p0 = self.p0.unsqueeze(1).expand_as(p)
        Gamma_0, iGi0 = self.Gamma_0.type(C128), self.iGi0.type(C128)
        Cqq = (Gamma_0 - torch.einsum('ij,jk,kl->il', Gamma_0, iGi0, Gamma_0)).unsqueeze(2).expand(-1, -1, n)  # (69)
        CQQ = Gt - torch.einsum('ijn,jk,lkn->iln', Gti, iGi0, Gti)                                             # (70)
        CqQ = torch.einsum('ij,jk,lkn->iln', Gamma_0, iGi0, Gti)                                               # (71)
        PIq = p0 - torch.einsum('ij,jk,kn->in', Gamma_0, iGi0, p0 - pi_i)                                      # (72)
        PIQ = pi_t + torch.einsum('ijn,jk,kn->in', Gti, iGi0, p0 - pi_i)                                       # (73)
        eps = (0.5 * torch.einsum('in,ijn,jn->n', b0, iA, b0)
               - 0.5 / hbar ** 2 * torch.einsum('in,ij,jn->n', p0 - pi_i, iGi0, p0 - pi_i))                    # (74)
        A = A / (2 * torch.sqrt(self.alpha * self.beta))
        detA = torch.det(A.permute(2, 0, 1))
        self.tracker.track("detA", detA)
        self.Cqq, self.CQQ, self.CqQ, self.PIq, self.PIQ, self.detA, self.eps = Cqq, CQQ, CqQ, PIq, PIQ, detA, eps
        G0 = Gamma_0.unsqueeze(2).expand(-1, -1, n)
        M = torch.einsum('ia,ijn,jb->abn', self.U, G0 + CQQ, self.U)                                           # (78)
        iM = torch.inverse(M.permute(2, 0, 1)).permute(1, 2, 0)
        detM = torch.det((M / (2 * np.pi)).permute(2, 0, 1))
        iM = torch.einsum('ai,ijn,bj->abn', self.U, iM, self.U)
        self.Rqq = Cqq - torch.einsum('ijn,jkn,lkn->iln', CqQ, iM, CqQ)                                        # (79)
        self.RQQ = G0 - torch.einsum('ij,jkn,kl->iln', Gamma_0, iM, Gamma_0)                                   # (80)
        self.RqQ = torch.einsum('ijn,jkn,kl->iln', CqQ, iM, Gamma_0)                                           # (81)
        self.Pq = PIq - torch.einsum('ijn,jkn,kn->in', CqQ, iM, PIQ - p0)                                      # (82)
        self.PQ = p0 + torch.einsum('ij,jkn,kn->in', Gamma_0, iM, PIQ - p0)                                    # (83)
        self.gamma = eps - 0.5 / hbar ** 2 * torch.einsum('in,ijn,jn->n', PIQ - p0, iM, PIQ - p0)              # (84)
        self.detM = detM
        self.tracker.track("detM", detM)

    def autocorrelation_qp(self):
        """eqn (85); propagators.py:1577-1614"""
        C, S = self.semiclassical_prefactor(), self.classical_action()
        q, p = self.initial_positions_and_momenta()
        Q, P = self.current_positions_and_momenta()
        q0 = self.q0.unsqueeze(1).expand_as(q).type(C128)
        pre = (self.detG0 ** (1 / 2) * self.detGt ** (1 / 4) * self.detGi ** (1 / 4)
               * 1 / torch.sqrt(self.detGi0) * C * torch.exp(1j / hbar * S)
               * 1 / torch.sqrt(self.detA) * self.tracker.signs("detA")
               * 1 / torch.sqrt(self.detM) * self.tracker.signs("detM"))
        return pre * torch.exp(
            self.gamma
            - 0.5 * torch.einsum('in,ijn,jn->n', q0 - q, self.Rqq, q0 - q)
            - 0.5 * torch.einsum('in,ijn,jn->n', q0 - Q, self.RQQ, q0 - Q)
            + torch.einsum('in,ijn,jn->n', q0 - q, self.RqQ, q0 - Q)
            - 1j / hbar * torch.einsum('in,in->n', self.Pq, q0 - q)
            + 1j / hbar * torch.einsum('in,in->n', self.PQ, q0 - Q))

    def ic_correlation(self, potential, energy0_es=0.0):
        """eqn (100); propagators.py:1652-1719"""
        cauto_qp = self.autocorrelation_qp()
        q, p = self.initial_positions_and_momenta()
        Q, P = self.current_positions_and_momenta()
        q0 = self.q0.unsqueeze(1).expand_as(q).type(C128)
        im = 1.0 / potential.masses()
        n1q = -hbar ** 2 * torch.einsum('k,kn->kn', im, potential.derivative_coupling_1st(q)).type(C128)
        n1Q = -hbar ** 2 * torch.einsum('k,kn->kn', im, potential.derivative_coupling_1st(Q)).type(C128)
        n2q = -hbar ** 2 * 0.5 * torch.einsum('k,kn->n', im, potential.derivative_coupling_2nd(q)).type(C128)
        n2Q = -hbar ** 2 * 0.5 * torch.einsum('k,kn->n', im, potential.derivative_coupling_2nd(Q)).type(C128)
        q, Q = q.type(C128), Q.type(C128)
        nacqQ = torch.einsum('in,ijn,jn->n', n1q, self.RqQ, n1Q)
        nacQ = n2Q + (torch.einsum('in,ijn,jn->n', q0 - Q, self.RQQ, n1Q)
                      - torch.einsum('in,ijn,jn->n', q0 - q, self.RqQ, n1Q)
                      - 1j / hbar * torch.einsum('in,in->n', self.PQ, n1Q))
        nacq = n2q + (torch.einsum('in,ijn,jn->n', q0 - q, self.Rqq, n1q)
                      - torch.einsum('in,jin,jn->n', q0 - Q, self.RqQ, n1q)
                      + 1j / hbar * torch.einsum('in,in->n', self.Pq, n1q))
        kic = (1.0 / hbar ** 2 * torch.exp(1j / hbar * self.t * torch.tensor(energy0_es))
               * (nacqQ + nacQ * nacq) * cauto_qp)
        return torch.sum(kic / self._mc_weight()).item()


def run_loop(propagator, potential, dt, nt, energy0_es=0.0):
    """the caller loop of cli.py:401-436: correlate, correlate, step -- nt times"""
    cauto = np.zeros(nt, dtype=complex)
    kic = np.zeros(nt, dtype=complex)
    for t in range(nt):
        cauto[t] = propagator.autocorrelation(energy0_es)
        kic[t] = propagator.ic_correlation(potential, energy0_es)
        propagator.step(potential, dt)
    return cauto, kic


# --------------------------------------------------------------------------
# sGDML force field: energy, gradient and analytic Hessian (gdml_predictor.py:96-250)
# --------------------------------------------------------------------------

class GDMLOracle(object):
    """E, dE/dr, d2E/drdr of an sGDML model for batches (B, 3N); restatement of GDMLPredict.forward.

    model keys used: sig, c, std, z, R_desc (Dd, M), R_d_desc_alpha (M, Dd), perms, tril_perms_lin
    (gdml_predictor.py:57-85).  The descriptor is x_d = 1/|r_i - r_j| over the pairs i > j in
    torch.tril_indices order.
    """

    def __init__(self, model):
        sig, self.c, self.std = int(model['sig']), float(model['c']), float(model.get('std', 1))
        self.q = np.sqrt(5) / sig
        self.n_atoms = int(np.asarray(model['z']).shape[0])
        desc = np.asarray(model['R_desc']).shape[0]
        n_perms = np.asarray(model['perms']).shape[0]
        perm = torch.tensor(np.asarray(model['tril_perms_lin'])).view(-1, n_perms).t()
        expand = lambda xs: xs.repeat(1, n_perms)[:, perm].reshape(-1, desc)
        self.xs_train = expand(torch.tensor(np.asarray(model['R_desc'], dtype=np.float64)).t())
        self.Jx_alphas = expand(torch.tensor(np.asarray(model['R_d_desc_alpha'], dtype=np.float64)))
        self.pair_i, self.pair_j = torch.tril_indices(self.n_atoms, self.n_atoms, offset=-1)

    def forward(self, r):
        N, q, A = self.n_atoms, self.q, self.Jx_alphas
        B = r.shape[0]
        k, l = self.pair_i, self.pair_j
        Dd = k.shape[0]
        pos = r.reshape(B, N, 3)
        diff = pos[:, k, :] - pos[:, l, :]                       # (B, Dd, 3)
        xs = 1.0 / diff.norm(dim=-1)                              # (B, Dd)
        xd = xs[:, None, :] - self.xs_train                       # (B, M, Dd)
        dist = xd.norm(dim=-1)                                    # (B, M)
        XA = torch.einsum('bmd,md->bm', xd, A)
        ef = 1.0 / 3.0 * q ** 4 * torch.exp(-q * dist)
        f = ef * (1.0 + q * dist) / q ** 2
        energy = torch.einsum('bm,bm->b', f, XA) * self.std + self.c
        # Jacobian of the descriptor: row d has -x^3 diff on atom k and +x^3 diff on atom l
        jac = torch.zeros(B, Dd, N, 3, dtype=r.dtype)
        idx = torch.arange(Dd)
        jd = -(xs ** 3)[:, :, None] * diff
        jac[:, idx, k, :] = jd
        jac[:, idx, l, :] -= jd
        jac = jac.reshape(B, Dd, 3 * N)
        gx = torch.einsum('bm,md->bd', f, A) - torch.einsum('bm,bmd->bd', ef * XA, xd)
        grad = torch.einsum('bd,bdx->bx', gx, jac) * self.std
        XJ = torch.einsum('bmd,bdx->bmx', xd, jac)
        AJ = torch.einsum('md,bdx->bmx', A, jac)
        JJ = torch.einsum('bdx,bdy->bxy', jac, jac)
        hess = torch.einsum('bm,bmx,bmy->bxy', ef * XA * q / dist, XJ, XJ)
        hess -= torch.einsum('bm,bxy->bxy', ef * XA, JJ)
        hess -= torch.einsum('bm,bmx,bmy->bxy', ef, AJ, XJ)
        hess -= torch.einsum('bm,bmx,bmy->bxy', ef, XJ, AJ)
        # second derivatives of the descriptor: T_d = 3 g x^5 diff diff^T - g x^3 1 on the (k,k), (l,l) blocks, -T_d on (k,l), (l,k)
        T = (3 * (gx * xs ** 5)[:, :, None, None] * diff[:, :, :, None] * diff[:, :, None, :]
             - (gx * xs ** 3)[:, :, None, None] * torch.eye(3))
        H4 = hess.reshape(B, N, 3, N, 3)
        for d in range(Dd):
            a, b = int(k[d]), int(l[d])
            H4[:, a, :, a, :] += T[:, d]
            H4[:, b, :, b, :] += T[:, d]
            H4[:, a, :, b, :] -= T[:, d]
            H4[:, b, :, a, :] -= T[:, d]
        return energy, grad, hess * self.std


class MolecularGDMLOracle(object):
    """potentials.py:641-744 with arrays instead of an fchk object"""

    def __init__(self, model, masses, nac0, origin=0.0):
        self.gdml = GDMLOracle(model)
        self._masses = torch.as_tensor(np.asarray(masses), dtype=torch.float64)
        self.nac0 = torch.as_tensor(np.asarray(nac0), dtype=torch.float64)
        self._dim = self._masses.shape[0]
        self._origin = float(origin)

    def dimensions(self):
        return self._dim

    def masses(self):
        return self._masses

    def harmonic_approximation(self, r):
        v, g, h = self.gdml.forward(r.permute(1, 0))
        return v - self._origin, g.permute(1, 0), h.permute(1, 2, 0)

    def derivative_coupling_1st(self, r):
        return self.nac0.unsqueeze(1).expand_as(r)

    def derivative_coupling_2nd(self, r):
        return torch.zeros_like(r)
