"""Atomic units and conversion factors.

Values are bit-identical to the reference (semiclassical/units.py:8-18): they
enter dt, the mode frequencies and the masses, so parity depends on them.
"""
hbar = 1.0

hartree_to_wavenumbers = 219474.63
hartree_to_ev = 27.211396132
bohr_to_angs = 0.529177249
autime_to_fs = 0.02418884326505
amu_to_aumass = 1822.888486192
