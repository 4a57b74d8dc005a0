"""CPU oracle of HermanKlukPropagator.coefficients() / norm() (reference propagators.py:657-686, 734-782).

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
