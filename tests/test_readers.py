"""Row N4 of SURVEY.md section 8f: the fchk reader without ASE, pinned to the REFERENCE reader's output.

tests/golden/readers_ref.npz was produced by the reference's `FormattedCheckpointFile` (tests/golden/
make_golden_driver.py) on the four fchk files the reference's own tests hold (copied as data into tests/golden/fchk/).
The second test is the reference's own known-answer test for the reader (tests/test_readers.py:21-46): frequencies from
the mass-weighted force constants equal the `Vib-E2` field of the coumarin files.
"""
import os

import numpy as np
import pytest

from semiclassical_amd import readers, units

HERE = os.path.dirname(os.path.abspath(__file__))
FCHK = os.path.join(HERE, "golden", "fchk")
REF = dict(np.load(os.path.join(HERE, "golden", "readers_ref.npz")))
NAMES = ["methylium_s0", "methylium_s1", "coumarin_s0", "coumarin_s1"]


def _open(name):
    with open(os.path.join(FCHK, name + ".fchk")) as f:
        return readers.FormattedCheckpointFile(f)


def _close(a, b, tol):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.max(np.abs(a - b)) <= tol * max(1.0, np.max(np.abs(b)))


@pytest.mark.parametrize("name", NAMES)
def test_reader_matches_reference_reader(name):
    fchk = _open(name)
    pos0, energy0, grad0, hess0 = fchk.harmonic_approximation()
    assert np.array_equal(pos0, REF[f"{name}_pos0"]) and np.array_equal(grad0, REF[f"{name}_grad0"])
    assert np.array_equal(hess0, REF[f"{name}_hess0"]) and float(energy0) == float(REF[f"{name}_energy0"])
    assert np.array_equal(fchk.masses(), REF[f"{name}_masses"])
    assert np.array_equal(np.asarray(fchk.atomic_numbers()), REF[f"{name}_atomic_numbers"])
    assert float(fchk.total_energy()) == float(REF[f"{name}_total_energy"])
    if f"{name}_nac" in REF:
        assert np.array_equal(fchk.nonadiabatic_coupling(), REF[f"{name}_nac"])
    # vibrational ground state: translations / rotations projected out with this package's own inertia code
    x0, Gamma_0, en_zpt = fchk.vibrational_groundstate()
    assert np.array_equal(x0, REF[f"{name}_x0"])
    assert _close(Gamma_0, REF[f"{name}_Gamma_0"], 1e-10)
    assert abs(en_zpt - float(REF[f"{name}_en_zpt"])) < 1e-12
    assert np.linalg.matrix_rank(Gamma_0, tol=1e-8) == len(x0) - 6


@pytest.mark.parametrize("name", ["coumarin_s0", "coumarin_s1"])
def test_frequencies_equal_vib_e2(name):
    """reference tests/test_readers.py:21-46"""
    fchk = _open(name)
    nmodes = fchk["Number of Normal Modes"]
    frequencies_fchk = fchk["Vib-E2"][:nmodes]
    assert np.array_equal(frequencies_fchk, REF[f"{name}_vib_e2"])
    masses = fchk.masses()
    pos, energy, grad, hess = fchk.harmonic_approximation()
    isqM = np.diag(1.0 / np.sqrt(masses))
    w2, _ = np.linalg.eigh(isqM @ hess @ isqM)
    assert np.isclose(w2[:6], np.zeros(6)).all()                 # translations and rotations
    frequencies = np.sqrt(w2[6:]) * units.hartree_to_wavenumbers
    assert np.isclose(frequencies, frequencies_fchk).all()


def test_nonadiabatic_coupling_is_readable():
    nac = _open("coumarin_s1").nonadiabatic_coupling()
    assert nac.shape == (51,) and np.any(nac != 0.0)
