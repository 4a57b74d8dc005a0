"""Small host-side linear algebra of the propagators (O(D^3), once per run).

Everything here is plain fp64 torch on the CPU: eigen-decompositions of the
D x D width matrices and the constants the HIP kernels consume.  It mirrors the
setup code of reference semiclassical/propagators.py (cited per function) and is
deliberately kept off the GPU so that results do not depend on the device
eigen-solver (eigenvector gauge).
"""
import math

import numpy as np
import torch

from .units import hbar

ZERO = 1.0e-8          # reference propagators.py:16
C128 = torch.complex128


def _eigh(A):
    return torch.linalg.eigh(A, UPLO='U')


def as_f64(x):
    return torch.as_tensor(x, dtype=torch.float64, device='cpu').detach().clone()


def sym_sqrtm(A):
    """A^{1/2} and pseudo-inverse A^{-1/2}, complex (D,D).  propagators.py:25-59"""
    w, V = _eigh(A)
    keep = abs(w) > ZERO
    wc, Vc = w.type(C128), V.type(C128)
    root = torch.einsum('ij,j,kj->ik', Vc, torch.sqrt(wc), Vc)
    iroot = torch.einsum('ij,j,kj->ik', Vc[:, keep], 1.0 / torch.sqrt(wc[keep]), Vc[:, keep])
    return root, iroot


def psd_sqrt_real(A):
    """real symmetric L with L L = A for a symmetric non-negative A (eigenvalues below zero by rounding are clamped)"""
    w, V = _eigh(A)
    return torch.einsum('ij,j,kj->ik', V, torch.sqrt(torch.clamp(w, min=0.0)), V)


def wavepacket_norm_factor(G):
    """(det'(G) / pi^rank)^(1/4), det' over the non-zero eigenvalues.  propagators.py:244-250, 284"""
    e, _ = _eigh(G)
    keep = abs(e) > ZERO
    return float((torch.prod(e[keep]) / np.pi ** int(torch.count_nonzero(keep))) ** 0.25)


def is_symmetric_non_negative(A, eps=1.0e-6):
    """propagators.py:61-82"""
    if torch.sum(abs(A - A.T)) / torch.sum(abs(A)) > eps:
        return False
    w, _ = _eigh(A)
    return bool((w >= -ZERO).all())


def is_diagonal(A):
    return bool((A - torch.diag(torch.diagonal(A)) == 0).all())


def sampling_matrices(Gamma_i, Gamma_0):
    """U, iGi0, iLz, detLz, d' of initial_conditions.  propagators.py:493-531"""
    wp, Vp = _eigh(Gamma_0 + Gamma_i)
    nzp = wp > ZERO
    U = Vp[:, nzp].type(C128)
    iGi0 = torch.einsum('ij,j,kj->ik', Vp[:, nzp], 1.0 / wp[nzp], Vp[:, nzp])
    iLp = torch.einsum('i,ji->ij', torch.sqrt(wp[nzp] / 2), Vp[:, nzp])
    wq, Vq = _eigh(Gamma_i @ iGi0 @ Gamma_0)
    nzq = wq > ZERO
    iLq = torch.einsum('i,ji->ij', 1.0 / torch.sqrt(2 * wq[nzq]), Vq[:, nzq])
    dprime = int(torch.count_nonzero(nzp))
    assert dprime == int(torch.count_nonzero(nzq)), \
        "number of non-zero modes for sampling of positions and momenta have to be the same"
    iLz = torch.block_diag(iLq, iLp)
    detLz = torch.prod(2 * torch.sqrt(wq[nzq] / wp[nzp]))
    return U, iGi0, iLz, detLz, dprime


class OverlapConstants(object):
    """constants of <q,p,Gi|q',p',Gj>.  propagators.py:125-179, 230"""

    def __init__(self, Gi, Gj):
        assert Gi.shape == Gj.shape, "width matrices Gi and Gj have to have the same shape"
        ei, _ = _eigh(Gi)
        ej, _ = _eigh(Gj)
        self.rank = int(torch.count_nonzero(abs(ei) > ZERO))
        assert self.rank == int(torch.count_nonzero(abs(ej) > ZERO)), \
            "Gi and Gj have to have the same rank and null space."
        detGi = torch.prod(ei[abs(ei) > ZERO])
        detGj = torch.prod(ej[abs(ej) > ZERO])
        eij, Vij = _eigh(Gi + Gj)
        keep = abs(eij) > ZERO
        self.B = torch.einsum('ij,j,kj->ik', Vij[:, keep], 1.0 / eij[keep], Vij[:, keep])   # (Gi+Gj)^+
        detGij = torch.prod(eij[keep])
        self.A = Gi @ self.B @ Gj
        self.C = Gj @ self.B
        self.fac = float(torch.sqrt(2.0 ** self.rank * torch.sqrt(detGi) * torch.sqrt(detGj) / detGij))
        self.diag = is_diagonal(self.A) and is_diagonal(self.B) and is_diagonal(self.C)


class PrefactorConstants(object):
    """constants of the HK prefactor matrix, eqn (29).  propagators.py:438-440, 969-994

    diag:   Gamma_i, Gamma_t diagonal with d' == D -- the sandwiches become elementwise scalings and the
            projection onto U (an orthogonal matrix when d' == D) leaves the determinant unchanged.
    dense:  L1 = U^T Gt^{1/2}, L2 = U^T Gt^{-1/2}, R1 = Gi^{-1/2} U, R2 = Gi^{1/2} U.
    """

    def __init__(self, Gamma_i, Gamma_t, U):
        D, dprime = U.shape
        self.dim, self.dprime = D, dprime
        sqGi, isqGi = sym_sqrtm(Gamma_i)
        sqGt, isqGt = sym_sqrtm(Gamma_t)
        self.sqGi, self.isqGi, self.sqGt, self.isqGt = sqGi, isqGi, sqGt, isqGt
        self.diag = (dprime == D and is_diagonal(Gamma_i) and is_diagonal(Gamma_t)
                     and bool((torch.diagonal(Gamma_i) > ZERO).all()) and bool((torch.diagonal(Gamma_t) > ZERO).all()))
        if self.diag:
            self.st = torch.sqrt(torch.diagonal(Gamma_t)).contiguous()
            self.si = torch.sqrt(torch.diagonal(Gamma_i)).contiguous()
        else:
            self.L1 = (U.T @ sqGt).contiguous()
            self.L2 = (U.T @ isqGt).contiguous()
            self.R1 = (isqGi @ U).contiguous()
            self.R2 = (sqGi @ U).contiguous()


class NacConstants(object):
    """constants of nacQ / nacq.  propagators.py:886-903 (constant coupling vector tau1, tau2 = 0)"""

    def __init__(self, Gamma_0, Gamma_i, iGi0, p0, masses, tau1, tau2_sum=0.0):
        n1 = -hbar ** 2 * tau1 / masses
        G = Gamma_0 @ iGi0
        self.rn = (G @ Gamma_i @ n1).contiguous()
        self.gn = (G.T @ n1).contiguous()
        self.p0n1 = float(torch.dot(p0, n1))
        self.n2 = float(-hbar ** 2 * 0.5 * tau2_sum)


def time_grid(nt, dt):
    """t_k of the propagator: t accumulates `t += dt` (propagators.py:655), not k*dt"""
    t, out = 0.0, np.empty(nt)
    for k in range(nt):
        out[k] = t
        t += dt
    return out


class WMConstants(object):
    """constants of the Walton-Manolopoulos prefactor, reference propagators.py:1102-1130, 1227-1238, 1264, 1295

    Everything is expressed in the projected space of dimension e = 2 d' (see csrc/sc_wm.hip).
    """

    def __init__(self, Gamma_0, Gamma_i, Gamma_t, iGi0, U, alpha, beta):
        D, dp = U.shape
        Ur = U.real.contiguous() if U.is_complex() else U
        pdet = lambda G, s: torch.prod((lambda e: e[abs(e) > ZERO] / s)(_eigh(G)[0]))
        detG0, detGi, detGt = pdet(Gamma_0, np.pi), pdet(Gamma_i, np.pi), pdet(Gamma_t, np.pi)
        detGi0 = pdet(Gamma_0 + Gamma_i, 2 * np.pi)
        self.pre = float(detG0 ** 0.5 * detGt ** 0.25 * detGi ** 0.25 / torch.sqrt(detGi0))     # :1598-1599
        self.pre_coef = float(detG0 ** 0.25 * detGt ** 0.25 * detGi ** 0.25 / torch.sqrt(detGi0))    # :1408-1409
        e0, V0 = _eigh(Gamma_0)
        keep = e0 > ZERO
        iGamma_0 = torch.einsum('ij,j,kj->ik', V0[:, keep], 1.0 / e0[keep], V0[:, keep])      # :1130
        self.dim, self.dprime = D, dp
        self.U = Ur.contiguous()
        self.Gt, self.G0, self.iGi0 = Gamma_t.contiguous(), Gamma_0.contiguous(), iGi0.contiguous()
        self.S = (iGi0 @ Gamma_0).contiguous()
        self.Cqq = (Gamma_0 - Gamma_0 @ iGi0 @ Gamma_0).contiguous()                          # (69)
        E = 2 * dp
        Cst = torch.zeros((E, E), dtype=C128)
        Cst[:dp, :dp] = 2 * alpha * (Ur.T @ Gamma_0 @ Ur) + Ur.T @ Gamma_i @ Ur
        Cst[dp:, dp:] = 2 * beta * (Ur.T @ iGamma_0 @ Ur)
        Cst[dp:, :dp] += -2j / hbar * (Ur.T @ Ur)
        self.Cst = Cst.contiguous()
        Bq = torch.zeros((D, E), dtype=C128)
        Bq[:, :dp] = Gamma_i @ Ur
        Bq[:, dp:] = -1j / hbar * Ur
        self.Bq = Bq.contiguous()
        self.inv_scale_a = 1.0 / (2.0 * math.sqrt(alpha * beta))
        self.inv_two_pi = 1.0 / (2.0 * np.pi)
