#!/usr/bin/env python
"""Exact quantum-mechanical IC correlation function of the 5-mode anharmonic AS model (chi = 0.02), the known-answer
fixture of the reference's own physics test (tests/test_propagators.py:419-426, 488-489).

Build container only: cuts the first 4 fs (the test uses 0 ... 3.75 fs) out of the reference's data file
tests/DATA/AnharmonicAS/5modes/ic_correlation_chi0.02_T0.001.dat (columns: time / fs, Re k_ic, Im k_ic) together with
the model file AS_model_chi0.02.dat / AS_model_chi0.00.dat (columns: omega / cm^-1, signed Huang-Rhys factor, NAC, chi).
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/tests/DATA/AnharmonicAS/5modes"


def main():
    ic = np.loadtxt(os.path.join(REF, "ic_correlation_chi0.02_T0.001.dat"))
    keep = ic[:, 0] <= 4.0
    np.savez_compressed(os.path.join(HERE, "qm_as5.npz"),
                        ic_chi002=ic[keep],
                        model_chi000=np.loadtxt(os.path.join(REF, "AS_model_chi0.00.dat")),
                        model_chi002=np.loadtxt(os.path.join(REF, "AS_model_chi0.02.dat")))
    print(int(keep.sum()), "rows kept")


if __name__ == "__main__":
    main()
