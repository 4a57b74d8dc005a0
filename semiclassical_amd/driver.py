"""`semi dynamics` / `semi rates` on the HIP engine: same JSON task keys and the same ``correlations.npz`` schema.

Mirrors reference semiclassical/cli.py:171-476 (``run_semiclassical_dynamics``) and :519-570
(``calculate_rates``): potential construction from the task, time grid (including quirk Q3: the stored
``times`` are ``linspace(0, nt*dt, nt)`` while the propagator advances by ``dt``), batching into repetitions,
trajectory-weighted running mean in the npz file, ``overwrite`` / accumulate semantics and the ``C(0) = 1``
assertion.  The time loop itself is ``propagator.run`` (no host synchronisation inside).

    python -m semiclassical_amd.driver dynamics input.json [--cuda ID]
    python -m semiclassical_amd.driver rates input.json

Not carried over: extxyz export and plotting (outside the hot path).
Potential types: "harmonic", "anharmonic AS", "gdml" (cli.py:178-303).
"""
import argparse
import json
import logging
import os

import numpy as np
import torch

from . import broadening, potentials, propagators, rates, readers, units
from .units import hbar

logger = logging.getLogger(__name__)


class ConfigurationError(Exception):
    pass


def _build_problem(task):
    """potential, q0, p0, Gamma_0, E_zpt, adiabatic gap from the task's 'potential' section (cli.py:178-303)"""
    p = task['potential']
    if p['type'] == "harmonic":
        with open(p['ground']) as f:
            freq_fchk = readers.FormattedCheckpointFile(f)
        with open(p['coupling']) as f:
            nacs_fchk = readers.FormattedCheckpointFile(f)
        with open(p['excited']) as f:
            excited_fchk = readers.FormattedCheckpointFile(f)
        potential = potentials.MolecularHarmonicPotential(freq_fchk, nacs_fchk)
        x0, Gamma_0, en_zpt = excited_fchk.vibrational_groundstate()
        q0 = torch.from_numpy(x0)
        p0 = torch.zeros_like(q0)
        Gamma_0 = torch.from_numpy(Gamma_0)
        potential.minimize(q0)
        gap = excited_fchk.total_energy() - potential.total_energy()
        return potential, q0, p0, Gamma_0, en_zpt, gap
    if p['type'] == "anharmonic AS":
        data = torch.from_numpy(np.loadtxt(p['model_file']))
        if len(data.shape) == 1:
            data = torch.reshape(data, (1, -1))
        omega = data[:, 0] / units.hartree_to_wavenumbers
        S, nac, chi = data[:, 1], data[:, 2], data[:, 3]
        dQ = torch.sqrt(2.0 * abs(S) / omega) * torch.sign(S)
        dQ[omega == 0.0] = 0.0
        potential = potentials.MorsePotential(omega, chi, nac)
        return potential, dQ, 0.0 * dQ, torch.diag(omega), torch.sum(hbar / 2.0 * omega).item(), np.nan
    if p['type'] == "gdml":
        from .gdml import MolecularGDMLPotential
        model_pot = np.load(p['ground'], allow_pickle=True)
        with open(p['coupling']) as f:
            nacs_fchk = readers.FormattedCheckpointFile(f)
        with open(p['excited']) as f:
            excited_fchk = readers.FormattedCheckpointFile(f)
        potential = MolecularGDMLPotential(model_pot, nacs_fchk)
        x0, Gamma_0, en_zpt = excited_fchk.vibrational_groundstate()
        q0 = torch.from_numpy(x0)
        p0 = torch.zeros_like(q0)
        Gamma_0 = torch.from_numpy(Gamma_0)
        potential.minimize(q0)          # raises the reference's RuntimeError when Newton + Armijo does not converge
        gap = excited_fchk.total_energy() - potential.total_energy()
        return potential, q0, p0, Gamma_0, en_zpt, gap
    raise ConfigurationError(f"Unknown potential type in {task['potential']}")


def run_semiclassical_dynamics(task, device='cuda'):
    torch.set_default_dtype(torch.float64)
    potential, q0, p0, Gamma_0, en_zpt, adiabatic_gap = _build_problem(task)
    Gamma_i = Gamma_t = Gamma_0

    dt = task['time_step_fs'] / units.autime_to_fs
    nt = task['num_steps']
    times = torch.linspace(0.0, nt * dt, nt)                                  # quirk Q3, cli.py:312-313

    batch_size = task.get('batch_size', 10000)
    num_trajectories = task.get('num_trajectories', 50000)
    num_repetitions = max(num_trajectories // batch_size, 1)
    num_samples = min(batch_size, num_trajectories)
    propagator_name = task.get('propagator', 'HK')

    filename = task['results'].get('correlations', 'correlations.npz')
    if task['results'].get('overwrite', True) is True or (not os.path.exists(filename)):
        np.savez(filename, propagator=propagator_name, times=times,
                 autocorrelation=np.zeros((nt,), dtype=complex), ic_correlation=np.zeros((nt,), dtype=complex),
                 adiabatic_gap=adiabatic_gap, zero_point_energy=en_zpt, trajectories=0)
    else:
        assert task.get('manual_seed', None) is None, \
            "Multiple runs with the same sequence of random numbers make no sense! Do not use `manual_seed` and `overwrite=False` at the same time"
        data = np.load(filename)
        assert np.array_equal(data['times'], times.numpy()), \
            f"Time steps in {filename} differ. Delete the old file or change the grid for time propagation."
        assert data['propagator'] == propagator_name, "Data produced with different propagators cannot be added."

    seed = task.get('manual_seed', None)
    if seed is not None:
        logger.warning("The random number generator should not be seeded manually unless for debugging!")
        torch.manual_seed(seed)

    for repetition in range(num_repetitions):
        logger.info(f"*** Repetition {repetition + 1} ***")
        if propagator_name == "WM":
            alpha = task.get('cell_width', 10000.0)
            propagator = propagators.WaltonManolopoulosPropagator(Gamma_i, Gamma_t, alpha, alpha, device=device)
        else:
            propagator = propagators.HermanKlukPropagator(Gamma_i, Gamma_t, device=device)
        propagator.initial_conditions(q0, p0, Gamma_0, ntraj=num_samples)
        calc_norm_every = task.get('calc_norm_every', 0)
        if calc_norm_every > 0:
            # convergence diagnostic of cli.py:424-429 (O(ntraj^2)): the fused loop is cut at the steps where the
            # norm is wanted
            parts = []
            for t0 in range(0, nt, calc_norm_every):
                logger.info(f" time/fs= {times[t0] * units.autime_to_fs}  norm= {propagator.norm():9.6f}")
                parts.append(propagator.run(potential, dt, min(calc_norm_every, nt - t0), energy0_es=en_zpt))
            autocorrelation_ = np.concatenate([c for c, _ in parts])
            ic_correlation_ = np.concatenate([k for _, k in parts])
        else:
            autocorrelation_, ic_correlation_ = propagator.run(potential, dt, nt, energy0_es=en_zpt)
        assert not np.isnan(autocorrelation_).any(), f"encountered NaN's in autocorrelation : {autocorrelation_}"
        assert not np.isnan(ic_correlation_).any(), f"encountered NaN's in IC correlation : {ic_correlation_}"

        data = dict(np.load(filename))
        ntraj_old, ntraj_new = data['trajectories'], num_samples
        ntraj_tot = ntraj_old + ntraj_new
        autocorrelation = (ntraj_new * autocorrelation_ + ntraj_old * data['autocorrelation']) / ntraj_tot
        ic_correlation = (ntraj_new * ic_correlation_ + ntraj_old * data['ic_correlation']) / ntraj_tot
        logger.info(f"<phi(0)|phi(0)>= {autocorrelation[0]}")
        assert abs(autocorrelation[0] - 1.0) < 1.0e-3
        data['trajectories'] = ntraj_tot
        data['autocorrelation'] = autocorrelation
        data['ic_correlation'] = ic_correlation
        data.pop('ic_rate', None)
        np.savez(filename, **data)


def calculate_rates(task):
    """Fourier transform of k_ic(t) into k_ic(E), cli.py:519-570 (including the factor 2 pi of :564)"""
    hwhmG = task.get('hwhmG_ev', 0.01)
    hwhmL = task.get('hwhmL_ev', 1.0e-6)
    sigma = hwhmG / np.sqrt(2.0 * np.log(2.0)) / units.hartree_to_ev
    gamma = hwhmL / units.hartree_to_ev
    broad = task.get('broadening', 'gaussian')
    if broad == "gaussian":
        lineshape = broadening.gaussian(sigma)
    elif broad == "lorentzian":
        lineshape = broadening.lorentzian(gamma)
    elif broad == "voigtian":
        lineshape = broadening.voigtian(sigma, gamma)
    else:
        raise ValueError("'broadening' should be one of 'gaussian', 'lorentzian' or 'voigtian'")
    corr_file = task.get('correlations', 'correlations.npz')
    rate_file = task.get('rates', 'correlations.npz')
    data = dict(np.load(corr_file))
    data['broadening'], data['hwhmG'], data['hwhmL'] = broad, hwhmG, hwhmL
    energies, ic_rate = rates.rate_from_correlation(data['times'], data['ic_correlation'], lineshape)
    ic_rate *= 2.0 * np.pi
    data['energies'] = energies[energies >= 0.0]
    data['ic_rate'] = ic_rate[energies >= 0.0].real
    np.savez(rate_file, **data)


def main(argv=None):
    logging.basicConfig(format="[%(module)-12s] %(message)s", level=logging.INFO)
    parser = argparse.ArgumentParser(prog="semi (semiclassical_amd)")
    sub = parser.add_subparsers(dest='command')
    dyn = sub.add_parser('dynamics', help="run semiclassical dynamics")
    dyn.add_argument('json_input', type=str, metavar='input.json')
    dyn.add_argument('--cuda', type=int, default=0, metavar='id')
    rat = sub.add_parser('rates', help="Fourier transform correlation functions into rates")
    rat.add_argument('json_input', type=str, metavar='input.json')
    args = parser.parse_args(argv)
    with open(args.json_input) as f:
        config = json.load(f)
    for task in config['semi']:
        if args.command == 'dynamics' and task['task'] == 'dynamics':
            run_semiclassical_dynamics(task, device=f"cuda:{args.cuda}")
        elif args.command == 'rates' and task['task'] == 'rates':
            calculate_rates(task)


if __name__ == "__main__":
    main()
