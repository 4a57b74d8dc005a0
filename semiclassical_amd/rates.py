"""Rate constants k(E) by Fourier transform of a damped correlation function.

Host-side NumPy, one inverse FFT of 2 nt - 1 points per run; follows
reference semiclassical/rates.py:20-82 including the cos^2 switching function and the unit conversion to s^-1.
"""
import numpy as np
from numpy import fft

from . import units

__all__ = ['rate_from_correlation']


def rate_from_correlation(times, correlation, lineshape):
    """k(E) = 1/(2 pi hbar) int dt exp(i E t/hbar) f(t) k(t) on the grid conjugate to ``times``.

    times: equidistant grid starting at 0; correlation: complex (nt,); lineshape: callable f(t).
    Returns (energies [Hartree], rate [s^-1]), both of length 2 nt - 1, sorted by energy.
    """
    assert times.min() == 0.0, "time grid `times` should start at 0.0"
    assert times.shape == correlation.shape, "arrays `times` and `correlation` should have the same length"
    nt = times.shape[0]
    t_max = times.max()
    m = 2 * nt - 1
    full_times = np.linspace(-t_max, t_max, m)
    energies = fft.fftfreq(m) * m / (2 * t_max) * 2.0 * np.pi
    full = np.zeros(m, dtype=complex)
    full[m // 2:] = correlation                       # t >= 0
    full[:m // 2] = correlation[1:].conj()[::-1]      # k(-t) = k(t)^*
    damp = np.cos(0.5 * np.pi * full_times / t_max) ** 2
    rate = 2 * t_max * fft.ifft(fft.ifftshift(damp * lineshape(full_times) * full))
    rate *= 1.0e15 / units.autime_to_fs
    return fft.fftshift(energies), fft.fftshift(rate)
