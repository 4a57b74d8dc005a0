"""A NumPy stand-in for the handful of ``ase`` calls the reference makes, for GOLDEN GENERATION ONLY.

The reference's ``readers.vibrational_groundstate`` and ``cli.run_semiclassical_dynamics`` import ``ase``
(absent from this image).  The golden-generating scripts in this directory install this
module as ``ase`` / ``ase.atoms`` / ``ase.io`` / ``ase.io.extxyz`` before importing the reference.  It is test
infrastructure written for this repository: the product package (semiclassical_amd.readers) has its own
inertia-tensor code and never imports it.
"""
import sys
import types

import numpy as np


class Atoms(object):
    """positions / masses container with the centre of mass and the principal axes of inertia"""

    def __init__(self, numbers=None):
        self.numbers = np.asarray(numbers, dtype=int)
        n = len(self.numbers)
        self._r = np.zeros((n, 3))
        self._m = np.ones(n)
        self._p = np.zeros((n, 3))

    def set_positions(self, r):
        self._r = np.array(r, dtype=float).reshape(-1, 3)

    def get_positions(self):
        return self._r.copy()

    def set_momenta(self, p):
        self._p = np.array(p, dtype=float).reshape(-1, 3)

    def set_masses(self, m):
        self._m = np.array(m, dtype=float)

    def get_masses(self):
        return self._m.copy()

    def get_center_of_mass(self):
        return self._m @ self._r / self._m.sum()

    def translate(self, shift):
        self._r = self._r + np.asarray(shift, dtype=float)

    def get_moments_of_inertia(self, vectors=False):
        r = self._r - self.get_center_of_mass()
        inertia = np.einsum('a,a,ij->ij', self._m, np.einsum('ai,ai->a', r, r), np.eye(3)) \
            - np.einsum('a,ai,aj->ij', self._m, r, r)
        evals, evecs = np.linalg.eigh(inertia)
        return (evals, evecs.T) if vectors else evals


def install():
    ase = types.ModuleType("ase")
    ase.__version__ = "stub (tests/golden/ase_stub.py)"
    atoms = types.ModuleType("ase.atoms")
    atoms.Atoms = Atoms
    io = types.ModuleType("ase.io")
    extxyz = types.ModuleType("ase.io.extxyz")
    extxyz.write_extxyz = lambda *a, **k: None
    io.extxyz = extxyz
    ase.atoms, ase.io, ase.Atoms = atoms, io, Atoms
    sys.modules.update({"ase": ase, "ase.atoms": atoms, "ase.io": io, "ase.io.extxyz": extxyz})
