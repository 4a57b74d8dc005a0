"""Lineshape functions in the time domain (reference semiclassical/broadening.py:25-147).

Each factory returns ``lineshape(t)``; its Fourier transform is a normalised distribution over energy.
``voigtian`` is the product form the reference documents; the reference's own implementation raises a
``TypeError`` (it calls ``gaussian(t, sigma)``, broadening.py:144) -- here it is g(t) l(t) with l(0) = 1.
``lorentzian`` keeps the reference's value l(0) = 0 (neither Heaviside branch includes t = 0, broadening.py:99-101).
"""
import numpy as np


def gaussian(sigma):
    def lineshape(t):
        return np.exp(-0.5 * sigma ** 2 * t ** 2) / (2.0 * np.pi)
    return lineshape


def lorentzian(gamma):
    def lineshape(t):
        l = 0.0 * t
        l[t > 0] = np.exp(-gamma * t[t > 0])
        l[t < 0] += np.exp(+gamma * t[t < 0])
        return l / (2.0 * np.pi)
    return lineshape


def voigtian(sigma, gamma):
    def lineshape(t):
        return np.exp(-0.5 * sigma ** 2 * t ** 2 - gamma * np.abs(t)) / (2.0 * np.pi)
    return lineshape
