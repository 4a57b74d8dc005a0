#!/usr/bin/env python
"""Generate the golden vectors in this directory by running the REFERENCE itself.

Run in the build container only (``/root/reference`` does not exist on the GPU
box):  ``python tests/golden/make_golden.py``.  The reference package is
imported from ``/root/reference`` with the two compatibility aliases that
torch >= 2 needs (``torch.symeig`` / ``torch.solve`` were removed; SURVEY.md
section 8c).  Nothing of the reference's source is stored -- only inputs and
outputs (arrays) end up in the ``.npz`` files.

What each file holds (all fp64 / complex128):
  inputs      potential parameters, q0, p0, Gamma_0/i/t, dt, E0, alpha/beta,
              zi (2D,n), probi (n,)              <- parity is pinned on these
  per step    c2 (nt+1,n) HK prefactor squared, [WM: detA, detM (nt+1,n)]
  snapshots   y (R,n) after steps SNAP, signs, per-trajectory cauto_qp
  outputs     C_auto[nt], k_ic[nt] (Python complex from the reference loop
              autocorrelation -> ic_correlation -> step, cli.py:401-436)
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "..", ".."))

torch.set_default_dtype(torch.float64)
torch.symeig = lambda A, eigenvectors=True, upper=True: tuple(torch.linalg.eigh(A, UPLO='U' if upper else 'L'))
torch.solve = lambda B, A: (torch.linalg.solve(A, B), None)

import logging
logging.disable(logging.CRITICAL)

from semiclassical.propagators import HermanKlukPropagator, WaltonManolopoulosPropagator  # noqa: E402
from semiclassical.potentials import (MorsePotential, NonHarmonicPotential,              # noqa: E402
                                      MolecularHarmonicPotential)
from semiclassical import units                                                         # noqa: E402


def run_case(name, make_propagator, potential, q0, p0, Gamma_0, dt, nt, E0, ntraj, snaps, extra,
             store_y=True):
    torch.manual_seed(0)
    prop = make_propagator()
    prop.initial_conditions(q0, p0, Gamma_0, ntraj=ntraj)
    is_wm = isinstance(prop, WaltonManolopoulosPropagator)
    out = dict(extra)
    out.update(q0=q0.numpy(), p0=p0.numpy(), Gamma_0=Gamma_0.numpy(),
               Gamma_i=prop.Gamma_i.numpy(), Gamma_t=prop.Gamma_t.numpy(),
               dt=float(dt), nt=nt, E0=float(E0), zi=prop.zi.numpy().copy(), probi=prop.probi.numpy().copy(),
               U=prop.U.numpy(), iGi0=prop.iGi0.numpy(), snaps=np.array(snaps))
    cauto = np.zeros(nt, dtype=complex)
    kic = np.zeros(nt, dtype=complex)
    c2 = np.zeros((nt + 1, ntraj), dtype=complex)
    detA = np.zeros((nt + 1, ntraj), dtype=complex)
    detM = np.zeros((nt + 1, ntraj), dtype=complex)

    def record(step):
        c2[step] = prop.sign_trackers["prefactorC"]["previous"].numpy()
        if is_wm:
            detA[step] = prop.detA.numpy()
            detM[step] = prop.detM.numpy()
        if step in snaps:
            if store_y:
                out[f"y_{step}"] = prop.y.numpy().copy()
            else:
                d = prop.dim
                out[f"qpS_{step}"] = np.vstack((prop.y[:2 * d].numpy(), prop.y[-1:].numpy()))
                out[f"ytraj0_{step}"] = prop.y[:, 0].numpy().copy()
            out[f"signs_{step}"] = prop.sign_trackers["prefactorC"]["signs"].numpy().copy()
            out[f"cauto_qp_{step}"] = prop.autocorrelation_qp().numpy().copy()
            if is_wm:
                out[f"signsA_{step}"] = prop.sign_trackers["detA"]["signs"].numpy().copy()
                out[f"signsM_{step}"] = prop.sign_trackers["detM"]["signs"].numpy().copy()
                out[f"gamma_{step}"] = prop.gamma.numpy().copy()

    record(0)
    for t in range(nt):
        cauto[t] = prop.autocorrelation(E0)
        kic[t] = prop.ic_correlation(potential, energy0_es=E0)
        prop.step(potential, dt)
        record(t + 1)
    out.update(cauto=cauto, kic=kic, c2=c2)
    if is_wm:
        out.update(detA=detA, detM=detM)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    flips = int((np.real(out[f"signs_{snaps[-1]}"]) < 0).sum())
    print(f"{name:24s} D={q0.shape[0]:3d} n={ntraj:4d} nt={nt:4d} C[-1]={cauto[-1]:.6f} "
          f"sign flips={flips}  {os.path.getsize(path) / 1024:.0f} KiB")


def as_model(omega_cm, S, nac, chi):
    omega = torch.as_tensor(omega_cm) / units.hartree_to_wavenumbers
    S, nac, chi = torch.as_tensor(S), torch.as_tensor(nac), torch.as_tensor(chi)
    dQ = torch.sqrt(2.0 * abs(S) / omega) * torch.sign(S)
    return omega, dQ, nac, chi


def synthetic_as60():
    """synthetic 60-mode anharmonic AS model"""
    rng = np.random.default_rng(60)
    omega_cm = np.linspace(160.0, 3300.0, 60)
    S = rng.uniform(0, 0.1, 60) * rng.choice([-1, 1], 60)
    nac = rng.normal(0, 1e-4, 60)
    chi = np.full(60, 0.02)
    return omega_cm, S, nac, chi


class _Fchk(object):
    """array-backed stand-in for the three accessor methods MolecularHarmonicPotential reads"""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def harmonic_approximation(self):
        return self.pos0, self.energy0, self.grad0, self.hess0

    def masses(self):
        return self._m

    def nonadiabatic_coupling(self):
        return self.nac0


def main():
    # ---- 1-D eps-Morse of Herman & Kluk 1986 (tests/test_propagators.py:116-328)
    nt = 100
    t_max = 12.0 / 40 * 2.0 * np.pi
    dt = np.linspace(0.0, t_max, nt)[1]
    pot = NonHarmonicPotential()
    q0, p0 = torch.tensor([7.3]), torch.tensor([0.0])
    Gi, G0 = torch.tensor([[5.0]]), torch.tensor([[1.0]])
    ex = dict(potential="nonharmonic", eps=pot.eps.numpy(), b=pot.b.numpy())
    run_case("hk_1d", lambda: HermanKlukPropagator(Gi, Gi), pot, q0, p0, G0, dt, nt, 0.5, 256,
             [1, 2, 10, nt], ex)
    run_case("wm_1d", lambda: WaltonManolopoulosPropagator(Gi, Gi, 100.0, 100.0), pot, q0, p0, G0, dt, nt, 0.5,
             128, [1, 2, 10, nt], dict(ex, alpha=100.0, beta=100.0))

    # ---- 5-mode adiabatic-shift model (tests/test_propagators.py:330-513)
    nt = 100
    t_max = 150.0 / units.autime_to_fs / 40.0
    dt = float(torch.linspace(0.0, t_max, nt)[1])
    for chi_val in (0.0, 0.02):
        data = np.loadtxt(f"{REF}/tests/DATA/AnharmonicAS/5modes/AS_model_chi{chi_val:.2f}.dat")
        omega, dQ, nac, chi = as_model(data[:, 0], data[:, 1], data[:, 2], data[:, 3])
        ex = dict(potential="morse", omega=omega.numpy(), chi=chi.numpy().copy(), nac=nac.numpy())
        pot = MorsePotential(omega, chi.clone(), nac)
        G = torch.diag(omega)
        E0 = torch.sum(0.5 * omega).item()
        tag = f"chi{int(round(chi_val * 100)):03d}"
        run_case(f"hk_as5_{tag}", lambda: HermanKlukPropagator(G, G), pot, dQ, 0.0 * dQ, G, dt, nt, E0, 256,
                 [1, 2, 10, nt], ex)
        if chi_val > 0:
            run_case(f"wm_as5_{tag}", lambda: WaltonManolopoulosPropagator(G, G, 500, 500), pot, dQ, 0.0 * dQ,
                     G, dt, nt, E0, 128, [1, 2, 10, nt], dict(ex, alpha=500.0, beta=500.0))

    # ---- synthetic 60-mode AS model (BASELINE.json config 2, parity subset)
    omega_cm, S, nac, chi = synthetic_as60()
    omega, dQ, nac, chi = as_model(omega_cm, S, nac, chi)
    ex = dict(potential="morse", omega=omega.numpy(), chi=chi.numpy().copy(), nac=nac.numpy())
    pot = MorsePotential(omega, chi.clone(), nac)
    G = torch.diag(omega)
    E0 = torch.sum(0.5 * omega).item()
    dt = 0.005 / units.autime_to_fs
    run_case("hk_as60", lambda: HermanKlukPropagator(G, G), pot, dQ, 0.0 * dQ, G, dt, 20, E0, 32,
             [1, 20], ex, store_y=False)
    # larger step so that the prefactor actually winds around the branch cut
    run_case("hk_as60_dt20", lambda: HermanKlukPropagator(G, G), pot, dQ, 0.0 * dQ, G, 20 * dt, 40, E0, 32,
             [1, 40], ex, store_y=False)

    # ---- methylium, Cartesian harmonic potential, rank-deficient Gamma (BASELINE.json config 3)
    from semiclassical_amd.readers import FormattedCheckpointFile
    d = f"{REF}/tests/DATA/examples/methylium_AH/"
    with open(d + "opt_freq_s0.fchk") as f:
        s0 = FormattedCheckpointFile(f)
    with open(d + "opt_freq_s1.fchk") as f:
        s1 = FormattedCheckpointFile(f)
    pos0, energy0, grad0, hess0 = s0.harmonic_approximation()
    x0, Gamma_0, en_zpt = s1.vibrational_groundstate()
    fchk = _Fchk(pos0=pos0, energy0=energy0, grad0=grad0, hess0=hess0, _m=s0.masses(),
                 nac0=s1.nonadiabatic_coupling())
    pot = MolecularHarmonicPotential(fchk, fchk)
    q0 = torch.from_numpy(x0)
    pot.minimize(q0)
    ex = dict(potential="harmonic", pos0=pos0, energy0=energy0, grad0=grad0, hess0=hess0, masses=s0.masses(),
              nac0=s1.nonadiabatic_coupling(), origin=pot._origin)
    G0 = torch.from_numpy(Gamma_0)
    dt = 0.02 / units.autime_to_fs
    run_case("hk_methylium", lambda: HermanKlukPropagator(G0, G0), pot, q0, 0.0 * q0, G0, dt, 60, en_zpt, 128,
             [1, 2, 10, 60], ex)
    run_case("wm_methylium", lambda: WaltonManolopoulosPropagator(G0, G0, 1.0e4, 1.0e4), pot, q0, 0.0 * q0, G0,
             dt, 40, en_zpt, 64, [1, 2, 10, 40], dict(ex, alpha=1.0e4, beta=1.0e4))


if __name__ == "__main__":
    main()


def rates_golden():
    """reference rates.rate_from_correlation on the golden k_ic(t) of hk_as5_chi002 (run after main())"""
    from semiclassical import rates as R, broadening as B
    g = dict(np.load(os.path.join(HERE, "hk_as5_chi002.npz")))
    nt, dt = int(g["nt"]), float(g["dt"])
    times = np.linspace(0.0, nt * dt, nt)
    sigma = 0.01 / np.sqrt(2.0 * np.log(2.0)) / units.hartree_to_ev
    en, rate = R.rate_from_correlation(times, g["kic"], B.gaussian(sigma))
    _, rate2 = R.rate_from_correlation(times, g["kic"], B.lorentzian(1e-3))
    np.savez_compressed(os.path.join(HERE, "rates_as5.npz"), times=times, kic=g["kic"], sigma=sigma, energies=en,
                        rate=rate, gamma=1e-3, rate_lorentzian=rate2)


if __name__ == "__main__":
    rates_golden()
