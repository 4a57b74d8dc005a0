"""Synthetic inputs for the BASELINE.json configurations the reference ships no data for.

  anharmonic_as_model(60)      config 2 / 4: 60-mode adiabatic-shift model (the reference only has a 5-mode file)
  sgdml_model(30, 200, 30)     config 5: an sGDML model of the shapes of a 30-atom molecule (the reference only has the
                               17-atom coumarin model)
Deterministic (fixed seeds), NumPy only; used by bench.py, tools/ and the tests.
"""
import numpy as np
import torch

from . import units


def anharmonic_as_model(dim=60):
    """omega, chi, nac, q0, dt of the synthetic `dim`-mode AS model: rng(60); omega = linspace(160, 3300) cm^-1;
    S = U(0, 0.1) x (+-1); nac ~ N(0, 1e-4); chi = 0.02; q0 = sign(S) sqrt(2|S|/omega); dt = 0.005 fs"""
    rng = np.random.default_rng(60)
    omega_cm = np.linspace(160.0, 3300.0, dim)
    S = rng.uniform(0, 0.1, dim) * rng.choice([-1, 1], dim)
    nac = rng.normal(0, 1e-4, dim)
    chi = np.full(dim, 0.02)
    omega = torch.from_numpy(omega_cm / units.hartree_to_wavenumbers)
    S, nac, chi = torch.from_numpy(S), torch.from_numpy(nac), torch.from_numpy(chi)
    q0 = torch.sqrt(2.0 * abs(S) / omega) * torch.sign(S)
    dt = 0.005 / units.autime_to_fs
    return omega, chi, nac, q0, dt


def sgdml_model(n_atoms, n_train, seed):
    """model dict (keys of an sGDML .npz: sig, c, std, z, R_desc (Dd, M), R_d_desc_alpha (M, Dd), perms,
    tril_perms_lin) and the geometry pos (n_atoms, 3) it was built around: atoms on a jittered lattice, training
    descriptors = descriptors of perturbed geometries, coefficients scaled like the coumarin model's"""
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


class ArrayFchk(object):
    """the three accessors MolecularGDMLPotential / MolecularHarmonicPotential take from an fchk object"""

    def __init__(self, masses, nac, atomic_numbers=None):
        self._masses, self._nac, self._z = np.asarray(masses), np.asarray(nac), atomic_numbers

    def masses(self):
        return self._masses

    def nonadiabatic_coupling(self):
        return self._nac

    def atomic_numbers(self):
        return self._z
