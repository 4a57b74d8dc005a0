"""rates / broadening: reference golden + the reference's own normalisation test."""
import numpy as np

from tests import cases


def test_rate_matches_reference_golden():
    from semiclassical_amd import broadening, rates
    g = cases.load("rates_as5")
    en, rate = rates.rate_from_correlation(g["times"], g["kic"], broadening.gaussian(float(g["sigma"])))
    assert np.array_equal(en, g["energies"])
    assert cases.rel_err(rate, g["rate"]) < 1e-13
    _, rate2 = rates.rate_from_correlation(g["times"], g["kic"], broadening.lorentzian(float(g["gamma"])))
    assert cases.rel_err(rate2, g["rate_lorentzian"]) < 1e-13


def test_lineshape_is_normalised():
    """reference tests/test_rates.py:15-47: the Fourier transform of g(t) integrates to 1"""
    from semiclassical_amd import broadening, rates, units
    sigma = 0.5 / np.sqrt(2.0 * np.log(2.0)) / units.hartree_to_ev
    times = np.linspace(0.0, 10.0, 500) / units.autime_to_fs
    for shape in (broadening.gaussian(sigma), broadening.voigtian(sigma, 1e-4)):
        w, G = rates.rate_from_correlation(times, np.ones_like(times), shape)
        G = G / (1.0e15 / units.autime_to_fs)
        assert abs(np.sum(G * (w[1] - w[0])) - 1.0) < 1e-6
