"""Formatted-checkpoint (fchk) input for molecular potentials.

One-off host-side I/O: it produces the arrays the
propagator path consumes -- (pos0, E0, grad0, hess0), masses, the NAC vector
and the vibrational ground state (x0, Gamma_0, E_zpt).  Behaviour follows
reference semiclassical/readers.py:29-396, but nothing here depends on ASE:
centre of mass and the inertia tensor are computed directly with NumPy.
"""
import re

import numpy as np
import scipy.linalg as sla

from . import units
from .units import hbar

__all__ = ["FormattedCheckpointFile"]

_HEADER = re.compile(r"^[A-Z]")


class FormattedCheckpointFile(object):
    """Dictionary-like view of a Gaussian 16 / QChem formatted checkpoint file.

    Scalars become Python numbers, ``N=`` arrays become 1-D ndarrays
    (reference readers.py:51-118).
    """

    def __init__(self, f):
        self.filename = getattr(f, "name", "<fchk>")
        self.data = {}
        field, kind, count, chunks = None, None, None, []

        def flush():
            if field is None or count is None:
                return
            text = " ".join(chunks)
            if kind is str:
                self.data[field] = text
                return
            arr = np.array(text.split(), dtype=kind)
            self.data[field] = arr if len(arr) == count else np.zeros(count, dtype=kind)

        for line in f.readlines():
            if not _HEADER.match(line):
                chunks.append(line.strip("\n"))
                continue
            flush()
            field, kind, count, chunks = None, None, None, []
            if len(line) < 43:
                continue                      # title / method lines
            kind = {"I": int, "R": float, "C": str}.get(line[43])
            if kind is None:
                continue
            field = line[:43].strip()
            if line[47:49] == "N=":
                count = int(line[49:])
            else:
                self.data[field] = kind(line[49:].strip()) if kind is not str else line[49:].strip()
        flush()

    def __getitem__(self, key):
        return self.data[key]

    def keys(self):
        return self.data.keys()

    # -- derived quantities ------------------------------------------------
    def atomic_numbers(self):
        return self.data["Atomic numbers"]

    def total_energy(self):
        return self.data["Total Energy"]

    def masses(self):
        """mass per Cartesian coordinate in electron masses (readers.py:365-376)"""
        return np.repeat(self.data["Real atomic weights"] * units.amu_to_aumass, 3)

    def nonadiabatic_coupling(self):
        return self.data["Nonadiabatic coupling"]

    def harmonic_approximation(self):
        """(pos, energy, grad, hess) in Cartesian coordinates (readers.py:144-190)"""
        nat = self.data["Number of atoms"]
        hess = np.zeros((3 * nat, 3 * nat))
        row, col = np.tril_indices(3 * nat)
        hess[row, col] = self.data["Cartesian Force Constants"]
        hess[col, row] = hess[row, col]
        return (self.data["Current cartesian coordinates"], np.array(self.data["Total Energy"]),
                self.data["Cartesian Gradient"], hess)

    def vibrational_groundstate(self):
        """(x0, Gamma_0, E_zpt) of the harmonic vibrational ground state.

        Translations and rotations are projected out of the mass-weighted
        Hessian before Gamma_0 = L.L^T is formed (readers.py:210-363).
        """
        x0, _, _, hess = self.harmonic_approximation()
        mass = self.masses()
        msq = np.sqrt(mass)
        hess_mwc = hess / np.outer(msq, msq)
        _, V = sla.eigh(hess_mwc)

        # centre-of-mass frame and principal axes of inertia
        m_at = mass[::3]
        r = x0.reshape(-1, 3)
        r = r - (m_at[:, None] * r).sum(0) / m_at.sum()
        inertia = np.zeros((3, 3))
        for mi, (x, y, z) in zip(m_at, r):
            inertia += mi * np.array([[y * y + z * z, -x * y, -x * z],
                                      [-x * y, x * x + z * z, -y * z],
                                      [-x * z, -y * z, x * x + y * y]])
        moments, axes = np.linalg.eigh(inertia)

        dim = len(mass)
        D = np.zeros((dim, dim))
        mwc = msq.reshape(-1, 3) * r
        for i in range(3):
            D[i::3, i] = msq[i::3]
        nz = 3
        for i in range(3):
            if moments[i] > 0.0:
                D[:, nz] = np.cross(axes[:, i], mwc).reshape(-1)
                nz += 1
        for i in range(nz):
            D[:, i] /= sla.norm(D[:, i])
        for n in range(nz, dim):
            D[:, n] = V[:, n]
            for m in range(n):
                D[:, n] -= np.dot(D[:, m], D[:, n]) * D[:, m]
            D[:, n] /= sla.norm(D[:, n])
        assert sla.norm(D.T @ D - np.eye(dim)) < 1.0e-10, "Gram-Schmidt orthogonalization failed"

        hess_int = D.T @ hess_mwc @ D
        wi2, Vi = sla.eigh(hess_int[nz:, nz:])
        wi = np.sqrt(wi2)
        en_zpt = 0.5 * hbar * np.sum(wi)
        L = hbar ** (-0.5) * msq[:, None] * (D[:, nz:] @ Vi) * np.sqrt(wi)[None, :]
        return x0, L @ L.T, en_zpt
