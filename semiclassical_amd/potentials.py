"""Potential energy surfaces behind the reference's potential protocol.

Same class names, constructor arguments and methods as reference
semiclassical/potentials.py (``dimensions``, ``masses``,
``harmonic_approximation``, ``derivative_coupling_1st/2nd``, ``minimize``,
``total_energy``).  In addition every class exposes ``_descriptor(device)``:
the plain parameter block (``sc_potential`` of include/semiclassical_hip.h) the
HIP step kernel evaluates V, grad V and hess V from -- the propagators never
call ``harmonic_approximation`` in the time loop and never materialise a
``(D, D, n)`` Hessian.

``harmonic_approximation`` itself is kept as a torch implementation because
callers outside the hot path use it (``minimize``, plotting scans).
"""
import logging

import numpy as np
import torch

from . import _lib
from ._lib import sc_potential, ptr

__all__ = ['NonHarmonicPotential', 'MorsePotential', 'MolecularHarmonicPotential']

logger = logging.getLogger(__name__)


class _DescriptorMixin(object):
    """caches the device copies of the parameter vectors per device; a cached block is reused only while the host
    parameters it was built from are unchanged (omega, chi, masses ... edited in place rebuild it)"""

    def _descriptor(self, device, dt=None):
        """``dt``: the time step the descriptor will be used with -- potentials with a constant Hessian attach the RK4
        step matrix of the monodromy equations for it (``sc_potential.lin_prop``)"""
        cache = self.__dict__.setdefault("_desc_cache", {})
        key = str(device)
        if dt is not None and self._has_step_matrix():
            base = self._descriptor(device)
            steps = cache[key][4]
            if dt not in steps:
                if len(steps) > 8:
                    steps.clear()
                phi = torch.from_numpy(self._step_matrix(dt)).contiguous().to(device)
                d = sc_potential.from_buffer_copy(base)
                d.lin_prop, d.lin_dt = ptr(phi), float(dt)
                steps[dt] = (d, phi)
            return steps[dt][0]
        kind, par0, par1, par2, scalar0 = self._parameters()
        host = [None if x is None else torch.as_tensor(x, dtype=torch.float64).detach().cpu()
                for x in (par0, par1, par2, self.masses())]
        stamp = (kind, float(scalar0))
        hit = cache.get(key)
        if hit is not None and hit[2] == stamp and all(
                (a is None and b is None) or (a is not None and b is not None and a.shape == b.shape and torch.equal(a, b))
                for a, b in zip(hit[3], host)):
            return hit[0]
        up = lambda x: None if x is None else x.contiguous().to(device)
        bufs = [up(host[0]), up(host[1]), up(host[2]), up(1.0 / host[3])]
        desc = sc_potential(kind=kind, dim=self.dimensions(), par0=ptr(bufs[0]), par1=ptr(bufs[1]),
                            par2=ptr(bufs[2]), scalar0=float(scalar0), inv_mass=ptr(bufs[3]))
        # bufs keeps the memory alive; the last entry holds the per-dt descriptors derived from this one
        cache[key] = (desc, bufs, stamp, [None if h is None else h.clone() for h in host], {})
        return desc

    def _step_matrix(self, dt):
        """None: the Hessian depends on the position.  Constant-Hessian potentials return Phi(dt) (2D x 2D)."""
        return None

    def _has_step_matrix(self):
        return False

    def _invalidate_descriptor(self):
        self.__dict__.pop("_desc_cache", None)


def _diag_hessian(d):
    dim, n = d.shape
    hess = torch.zeros((dim, dim, n), dtype=d.dtype, device=d.device)
    torch.diagonal(hess, dim1=0, dim2=1)[...] = d.transpose(0, 1)
    return hess


class NonHarmonicPotential(_DescriptorMixin):
    """eps*Morse + (1-eps)*harmonic oscillator of Herman & Kluk (1986), eqn (7).

    V(x) = eps/(2 b^2) (1 - exp(-b x))^2 + (1-eps) x^2/2      reference potentials.py:25-204
    """

    def __init__(self, eps=torch.tensor([0.975]), b=torch.tensor([(12.0) ** (-0.5)])):
        self.eps = eps
        self.b = b

    def dimensions(self):
        return self.eps.size()[0]

    def masses(self):
        return torch.ones(self.dimensions())

    def _parameters(self):
        return _lib.SC_POT_EPS_MORSE, self.eps, self.b, None, 0.0

    def harmonic_approximation(self, r):
        eps = self.eps.to(r.device).unsqueeze(1).expand_as(r)
        b = self.b.to(r.device).unsqueeze(1).expand_as(r)
        e1, e2 = torch.exp(-b * r), torch.exp(-2 * b * r)
        vpot = torch.sum(eps / (2 * b ** 2) * (1.0 - e1) ** 2 + (1 - eps) * 0.5 * r ** 2, 0)
        grad = eps / b * (e1 - e2) + (1 - eps) * r
        return vpot, grad, _diag_hessian(eps * (2 * e2 - e1) + (1 - eps))

    def derivative_coupling_1st(self, r):
        return torch.ones_like(r)

    def derivative_coupling_2nd(self, r):
        return torch.zeros_like(r)


class MorsePotential(_DescriptorMixin):
    """Morse potential with anharmonicity chi, V = D (1 - exp(-a r))^2 per mode.

    a = sqrt(2 omega chi), D = omega / (4 chi); all chi == 0 selects the harmonic
    limit V = omega^2 r^2 / 2.  reference potentials.py:208-397
    """

    def __init__(self, omega, chi, nac):
        self.omega = omega
        self.nac = nac
        if (chi == 0.0).all():
            logger.info("Potential is harmonic.")
        else:
            # harmonic modes get a tiny anharmonicity, in place, as in the reference (potentials.py:250)
            chi[chi == 0.0] += 1.0e-4
        self.chi = chi
        self.a = torch.sqrt(2 * omega * chi)
        self.D = 0.25 * omega / chi

    def dimensions(self):
        return self.a.size()[0]

    def masses(self):
        return torch.ones(self.dimensions())

    def _is_harmonic(self):
        return bool((self.chi == 0.0).all())

    def _parameters(self):
        if self._is_harmonic():
            return _lib.SC_POT_HARMONIC_SEP, self.omega ** 2, None, None, 0.0
        return _lib.SC_POT_MORSE, self.a, self.D, None, 0.0

    def harmonic_approximation(self, r):
        col = lambda v: v.to(r.device).unsqueeze(1).expand_as(r)
        if self._is_harmonic():
            w2 = col(self.omega) ** 2
            return torch.sum(0.5 * w2 * r ** 2, 0), w2 * r, _diag_hessian(w2.clone())
        a, D = col(self.a), col(self.D)
        e = torch.exp(-a * r)
        vpot = torch.sum(D * (1.0 - e) ** 2, 0)
        grad = 2 * a * D * e * (1.0 - e)
        return vpot, grad, _diag_hessian(2 * a ** 2 * D * e * (2 * e - 1.0))

    def derivative_coupling_1st(self, r):
        return self.nac.to(r.device).unsqueeze(1).expand_as(r)

    def derivative_coupling_2nd(self, r):
        return torch.zeros_like(r)


class _MolecularPotentialBase(_DescriptorMixin):
    _masses, _dim = None, None
    _origin = 0.0

    def dimensions(self):
        return self._dim

    def masses(self):
        return self._masses

    def total_energy(self):
        return self._origin

    def minimize(self, r_guess, maxiter=200, rtol=1.0e-5, gtol=1.0e-7):
        """Newton steps with Armijo backtracking to the local minimum; afterwards energies are
        measured from the minimum.  One-off host-side setup.  reference potentials.py:435-526"""
        self._origin = 0.0
        self._invalidate_descriptor()
        r = r_guess.unsqueeze(1)
        for i in range(maxiter):
            energy, grad, hess = self.harmonic_approximation(r)
            dr = torch.linalg.solve(hess.squeeze(), -grad)
            slope = torch.sum(grad * dr)
            if slope > 0.0:
                dr = -grad
                slope = torch.sum(grad * dr)
            if torch.norm(grad) < gtol or torch.norm(dr) < rtol:
                break
            step, rho, c = 1.0, 0.3, 1.0e-4
            for _ in range(100):
                r_new = r + step * dr
                e_new, _, _ = self.harmonic_approximation(r_new)
                if e_new <= energy + c * step * slope:
                    break
                step *= rho
            else:
                raise RuntimeError("Linesearch failed! Could not find a step length that satisfies the "
                                   "sufficient decrease condition.")
            r = r_new
        else:
            raise RuntimeError(f"Could not find minimum within {maxiter} iterations.")
        emin, _, _ = self.harmonic_approximation(r)
        self._origin = emin.item()
        self._invalidate_descriptor()


class MolecularHarmonicPotential(_MolecularPotentialBase):
    """Second-order expansion around a reference geometry, Cartesian coordinates.

    Built from two fchk-like objects exactly as in the reference (potentials.py:529-638); any object
    with ``harmonic_approximation()``, ``masses()`` and ``nonadiabatic_coupling()`` works, e.g.
    ``semiclassical_amd.readers.FormattedCheckpointFile``.
    """

    def __init__(self, freq_fchk, nac_fchk):
        self.pos0, self.energy0, self.grad0, self.hess0 = (
            torch.from_numpy(np.asarray(x, dtype=np.float64)) for x in freq_fchk.harmonic_approximation())
        self.nac0 = torch.from_numpy(np.asarray(nac_fchk.nonadiabatic_coupling(), dtype=np.float64))
        self._masses = torch.from_numpy(np.asarray(freq_fchk.masses(), dtype=np.float64))
        self._dim = len(self._masses)

    @classmethod
    def from_arrays(cls, pos0, energy0, grad0, hess0, masses, nac0, origin=0.0):
        class _Arrays(object):
            def harmonic_approximation(self_):
                return pos0, energy0, grad0, hess0

            def masses(self_):
                return masses

            def nonadiabatic_coupling(self_):
                return nac0
        pot = cls(_Arrays(), _Arrays())
        pot._origin = float(origin)
        return pot

    def _parameters(self):
        return (_lib.SC_POT_HARMONIC_DENSE, self.pos0, self.grad0, self.hess0,
                float(self.energy0) - self._origin)

    def _step_matrix(self, dt):
        """RK4 of the linear system d/dt [X; Y] = G [X; Y], G = [[0, 1/m], [-hess0, 0]] (the monodromy equations of
        reference propagators.py:342-357 with a constant Hessian) is the product with
        Phi = 1 + hG + (hG)^2/2 + (hG)^3/6 + (hG)^4/24: the four stages of propagators.py:86-119 written out"""
        D = self._dim
        if not self._has_step_matrix():
            return None              # the kernel that takes Phi holds D <= 16
        G = np.zeros((2 * D, 2 * D), dtype=np.longdouble)
        G[:D, D:] = np.diag(1.0 / self._masses.numpy().astype(np.longdouble))
        G[D:, :D] = -self.hess0.numpy().astype(np.longdouble)
        hG, one = np.longdouble(dt) * G, np.eye(2 * D, dtype=np.longdouble)
        phi = one + hG @ (one + hG @ (one + hG @ (one + hG / 4) / 3) / 2)
        return np.ascontiguousarray(phi.astype(np.float64))

    def _has_step_matrix(self):
        return self._dim <= 16

    def _normal_modes(self, dt):
        """RK4 of the monodromy equations in normal-mode coordinates (``sc_hk_run_modal``, include/semiclassical_hip.h): with
        W = m^-1/2 hess0 m^-1/2 = U diag(lam) U^T the step matrix Phi(dt) of ``_step_matrix`` is T diag-by-mode(phi_a) T^-1,
        T = blockdiag(A, B), A = m^-1/2 U, B = m^1/2 U (the RK4 polynomial commutes with the change of basis).
        Returns (A, B, A^-1, B^-1, phi[D][4] = (phi_qq, phi_qp, phi_pq, phi_pp) per mode) as float64 arrays."""
        m = self._masses.numpy().astype(np.float64)
        h = self.hess0.numpy().astype(np.float64)
        sm = np.sqrt(m)
        lam, U = np.linalg.eigh(0.5 * (h + h.T) / np.outer(sm, sm))
        A, B = U / sm[:, None], U * sm[:, None]
        Ainv, Binv = U.T * sm[None, :], U.T / sm[None, :]
        phi = np.zeros((self._dim, 4))
        one = np.eye(2, dtype=np.longdouble)
        for a in range(self._dim):
            hG = np.longdouble(dt) * np.array([[0.0, 1.0], [-lam[a], 0.0]], dtype=np.longdouble)
            P = one + hG @ (one + hG @ (one + hG @ (one + hG / 4) / 3) / 2)
            phi[a] = [P[0, 0], P[0, 1], P[1, 0], P[1, 1]]
        return A, B, Ainv, Binv, phi

    def harmonic_approximation(self, r):
        dev = r.device
        pos0, grad0, hess0 = self.pos0.to(dev), self.grad0.to(dev), self.hess0.to(dev)
        dim, n = r.size()
        dr = r - pos0.unsqueeze(1).expand_as(r)
        vpot = (self.energy0.to(dev) + torch.einsum('in,i->n', dr, grad0)
                + 0.5 * torch.einsum('in,ij,jn->n', dr, hess0, dr))
        grad = grad0.unsqueeze(1).expand_as(r) + torch.einsum('ij,jn->in', hess0, dr)
        return vpot - self._origin, grad, hess0.unsqueeze(2).expand(-1, -1, n)

    def derivative_coupling_1st(self, r):
        return self.nac0.to(r.device).unsqueeze(1).expand_as(r)

    def derivative_coupling_2nd(self, r):
        return torch.zeros_like(r)
