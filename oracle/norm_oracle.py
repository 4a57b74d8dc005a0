"""CPU oracle of HermanKlukPropagator.coefficients() / norm() / wavefunction()
(reference propagators.py:657-686, 734-782, 688-732 with CoherentStatesWavefunction :243-292).

TEST INFRASTRUCTURE ONLY (see oracle/sc_oracle.py).  Pinned by tests/golden/hk_norms.npz, produced by the reference.
"""
import numpy as np
import torch

from . import sc_oracle as orc

hbar = orc.hbar


def coefficients(prop):
    """v_i = C e^{iS/hbar} <qi,pi|phi0> / ((2 pi hbar)^d n P_i)"""
    qi, pi = prop.initial_positions_and_momenta()
    v0 = prop.csoi0(qi, pi, prop.q0, prop.p0).squeeze()
    v = prop.semiclassical_prefactor() * torch.exp(1j / hbar * prop.classical_action()) / (2 * np.pi * hbar) ** prop.dim * v0
    return v / (prop.ntraj * prop.probi)


def norm(prop, chunk=1000):
    """|psi| = sqrt(sum_ij v_i^* <q_i,p_i,Gt|q_j,p_j,Gt> v_j), accumulated block by block"""
    v = coefficients(prop)
    q, p = prop.current_positions_and_momenta()
    nchunk = prop.ntraj // chunk + 1
    total = torch.tensor([0.0j])
    for qi, pi, vi in zip(torch.chunk(q, nchunk, dim=1), torch.chunk(p, nchunk, dim=1), torch.chunk(v, nchunk, dim=0)):
        for qj, pj, vj in zip(torch.chunk(q, nchunk, dim=1), torch.chunk(p, nchunk, dim=1), torch.chunk(v, nchunk, dim=0)):
            total += torch.einsum('i,ij,j', vi.conj(), prop.csott(qi, pi, qj, pj), vj)
    return torch.sqrt(total.real).item()


def wavefunction(prop, x):
    """psi(x_k) = sum_n v_n (det'Gt / pi^rank)^(1/4) exp(-1/2 (x-q_n)^T Gt (x-q_n) + i/hbar p_n.(x-q_n)),  x (dim,nx)
    (reference propagators.py:274-292, 722-732; the chunking over the grid does not change the sums)"""
    x = torch.as_tensor(x, dtype=torch.float64)
    Gt = prop.Gamma_t
    e = torch.linalg.eigvalsh(Gt)
    keep = abs(e) > orc.ZERO
    fac = (torch.prod(e[keep]) / np.pi ** int(torch.count_nonzero(keep))) ** 0.25
    q, p = prop.current_positions_and_momenta()
    v = coefficients(prop)
    dx = x.unsqueeze(1) - q.unsqueeze(2)                                   # (dim, ntraj, nx)
    expo = -0.5 * torch.einsum('inx,ij,jnx->nx', dx, Gt, dx) + 1j / hbar * torch.einsum('in,inx->nx', p, dx)
    return torch.sum(v.unsqueeze(1) * fac * torch.exp(expo), 0).numpy()


def wm_coefficients(prop):
    """Walton-Manolopoulos coefficients of eqn (75), x-independent part (reference propagators.py:1391-1432)"""
    d, n = prop.dim, prop.ntraj
    C, S = prop.semiclassical_prefactor(), prop.classical_action()
    v = (prop.detG0 ** 0.25 * prop.detGt ** 0.25 * prop.detGi ** 0.25 / torch.sqrt(prop.detGi0) / (2 * np.pi) ** d
         * C * torch.exp(1j / hbar * S) / torch.sqrt(prop.detA) * prop.tracker.signs("detA") * torch.exp(prop.eps))
    q, p = prop.initial_positions_and_momenta()
    dq = (prop.q0.unsqueeze(1) - q).type(torch.complex128)
    v = v * torch.exp(-0.5 * torch.einsum('in,ijn,jn->n', dq, prop.Cqq, dq)
                      - 1j / hbar * torch.einsum('in,in->n', prop.PIq, dq))
    return v / (n * prop.probi)


def wm_wavefunction(prop, x):
    """WM wavefunction on the grid x (dim,nx) (reference propagators.py:1434-1482)"""
    x = torch.as_tensor(x, dtype=torch.float64)
    v = wm_coefficients(prop)
    q, _ = prop.initial_positions_and_momenta()
    Q, _ = prop.current_positions_and_momenta()
    dq = (prop.q0.unsqueeze(1) - q).type(torch.complex128)
    dx = (x.unsqueeze(1) - Q.unsqueeze(2)).type(torch.complex128)                      # (dim, ntraj, nx)
    expo = (-0.5 * torch.einsum('inx,ijn,jnx->nx', dx, prop.CQQ, dx) + torch.einsum('in,ijn,jnx->nx', dq, prop.CqQ, dx)
            + 1j / hbar * torch.einsum('in,inx->nx', prop.PIQ, dx))
    return torch.sum(v.unsqueeze(1) * torch.exp(expo), 0).numpy()


def wm_norm(prop, chunk=64):
    """norm of the WM wavefunction (reference propagators.py:1484-1575), block by block over the trajectory pairs"""
    d, n = prop.dim, prop.ntraj
    v = wm_coefficients(prop)
    q, _ = prop.initial_positions_and_momenta()
    Q, _ = prop.current_positions_and_momenta()
    q0 = prop.q0.unsqueeze(1).expand_as(q).type(torch.complex128)
    dvec = torch.einsum('ban,bn->an', prop.CqQ, q0 - q) + 1j / hbar * prop.PIQ
    U = prop.U.type(torch.complex128)
    nchunk = n // chunk + 1
    total = torch.tensor([0.0j])
    for Qi, di, Ci, vi in zip(torch.chunk(Q, nchunk, dim=1), torch.chunk(dvec, nchunk, dim=1),
                              torch.chunk(prop.CQQ, nchunk, dim=2), torch.chunk(v, nchunk, dim=0)):
        ni = vi.shape[0]
        for Qj, dj, Cj, vj in zip(torch.chunk(Q, nchunk, dim=1), torch.chunk(dvec, nchunk, dim=1),
                                  torch.chunk(prop.CQQ, nchunk, dim=2), torch.chunk(v, nchunk, dim=0)):
            nj = vj.shape[0]
            dQ = (Qj.unsqueeze(1).expand(-1, ni, -1) - Qi.unsqueeze(2).expand(-1, -1, nj)).type(torch.complex128)
            di_, dj_ = di.unsqueeze(2).expand(-1, -1, nj), dj.unsqueeze(1).expand(-1, ni, -1)
            Cj_ = Cj.unsqueeze(2).expand(-1, -1, ni, -1)
            Dij = Ci.unsqueeze(3).expand(-1, -1, -1, nj).conj() + Cj_
            Dp = torch.einsum('ia,ijmn,jb->abmn', U, Dij, U).permute(2, 3, 0, 1)
            iD = torch.einsum('ai,ijmn,bj->abmn', U, torch.inverse(Dp).permute(2, 3, 0, 1), U)
            detD = torch.det(Dp / (2 * np.pi))
            bij = torch.einsum('abij,bij->aij', Cj_, dQ) + di_.conj() + dj_
            olap = 1 / torch.sqrt(detD) * torch.exp(-0.5 * torch.einsum('aij,abij,bij->ij', dQ, Cj_, dQ)
                                                    - torch.einsum('aij,aij->ij', dj_, dQ)
                                                    + 0.5 * torch.einsum('aij,abij,bij->ij', bij, iD, bij))
            total += torch.einsum('i,ij,j', vi.conj(), olap, vj)
    return torch.sqrt(total.real).item()
