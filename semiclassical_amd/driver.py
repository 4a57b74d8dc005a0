"""`semi dynamics` / `semi rates` tasks on the HIP engine.

Drop-in for the JSON task lists of the reference (README.rst:90-102; semiclassical/cli.py:171-476
``run_semiclassical_dynamics``, :519-570 ``calculate_rates``): the same task keys with the same defaults, the same
``correlations.npz`` keys, the same accumulate / overwrite semantics and the same error texts -- those are the
contract.  The code around the contract is organised differently from the reference:

    ProblemSetup       what a potential section resolves to (one builder per potential type, registered by name)
    CorrelationStore   the npz file as a running, trajectory-weighted mean over repetitions and over separate runs
    propagate_batch    one repetition on the device; the time loop is ``propagator.run`` (no host sync inside)

    python -m semiclassical_amd.driver dynamics input.json [--cuda ID]
    python -m semiclassical_amd.driver rates input.json

Not carried over: extxyz export and plotting (outside the hot path).
"""
import argparse
import json
import logging
import os
from collections import namedtuple

import numpy as np
import torch

from . import broadening, potentials, propagators, rates, readers, units
from .units import hbar

logger = logging.getLogger(__name__)


class ConfigurationError(Exception):
    pass


# ---------------------------------------------------------------------------------------------------------------------
# the 'potential' section of a task
# ---------------------------------------------------------------------------------------------------------------------

ProblemSetup = namedtuple("ProblemSetup", "potential q0 p0 Gamma_0 zero_point_energy adiabatic_gap")

_SETUPS = {}


def _potential_type(name):
    def register(builder):
        _SETUPS[name] = builder
        return builder
    return register


def _fchk(path):
    with open(path) as handle:
        return readers.FormattedCheckpointFile(handle)


def _molecular_setup(surface, excited_state):
    """Final-state surface + initial wavepacket of a molecule: the wavepacket is the vibrational ground state of the
    excited-state fchk file; the surface is relaxed from there and its minimum becomes the energy origin, which fixes
    the adiabatic gap (cli.py:187-197, 293-300)."""
    centre, widths, zero_point = excited_state.vibrational_groundstate()
    q0 = torch.from_numpy(centre)
    surface.minimize(q0)
    gap = excited_state.total_energy() - surface.total_energy()
    logger.info(f"  adiabatic excitation energy               : {gap * units.hartree_to_ev:.4f} eV")
    return ProblemSetup(surface, q0, torch.zeros_like(q0), torch.from_numpy(widths), zero_point, gap)


@_potential_type("harmonic")
def _setup_harmonic(section):
    surface = potentials.MolecularHarmonicPotential(_fchk(section['ground']), _fchk(section['coupling']))
    return _molecular_setup(surface, _fchk(section['excited']))


@_potential_type("gdml")
def _setup_gdml(section):
    from .gdml import MolecularGDMLPotential
    model = np.load(section['ground'], allow_pickle=True)
    surface = MolecularGDMLPotential(model, _fchk(section['coupling']))
    return _molecular_setup(surface, _fchk(section['excited']))     # minimize() may raise, as the reference's does


@_potential_type("anharmonic AS")
def _setup_adiabatic_shift(section):
    """model file with one row per mode: frequency / cm^-1, Huang-Rhys factor (its sign is the direction of the
    displacement), coupling, anharmonicity (README.rst:367-373, cli.py:229-285)"""
    table = torch.from_numpy(np.atleast_2d(np.loadtxt(section['model_file'])))
    logger.info("vibrational modes (cm^-1):")
    logger.info(table[:, 0])
    omega = table[:, 0] / units.hartree_to_wavenumbers
    huang_rhys, coupling, chi = table[:, 1], table[:, 2], table[:, 3]
    shift = torch.sign(huang_rhys) * torch.sqrt(2.0 * abs(huang_rhys) / omega)       # S = 1/2 dQ^2 omega
    shift[omega == 0.0] = 0.0                                                        # zero modes are not displaced
    zero_point = torch.sum(hbar / 2.0 * omega).item()
    return ProblemSetup(potentials.MorsePotential(omega, chi, coupling), shift, 0.0 * shift, torch.diag(omega),
                        zero_point, np.nan)


def build_problem(task):
    section = task['potential']
    builder = _SETUPS.get(section['type'])
    if builder is None:
        raise ConfigurationError(f"Unknown potential type in {task['potential']}")
    return builder(section)


# ---------------------------------------------------------------------------------------------------------------------
# correlations.npz
# ---------------------------------------------------------------------------------------------------------------------

class CorrelationStore(object):
    """The result file: ``propagator, times, autocorrelation, ic_correlation, adiabatic_gap, zero_point_energy,
    trajectories`` (cli.py:346-353), re-read and re-written after every batch so that an interrupted run keeps what it
    has (cli.py:453-476)."""

    def __init__(self, path):
        self.path = path

    def start(self, task, propagator_name, times, setup):
        fresh = task['results'].get('overwrite', True) is True or not os.path.exists(self.path)
        if fresh:
            nt = len(times)
            np.savez(self.path, propagator=propagator_name, times=times,
                     autocorrelation=np.zeros((nt,), dtype=complex), ic_correlation=np.zeros((nt,), dtype=complex),
                     adiabatic_gap=setup.adiabatic_gap, zero_point_energy=setup.zero_point_energy, trajectories=0)
            return
        # adding to the results of an earlier run
        assert task.get('manual_seed', None) is None, \
            "Multiple runs with the same sequence of random numbers make no sense! Do not use `manual_seed` and `overwrite=False` at the same time"
        previous = np.load(self.path)
        assert np.array_equal(previous['times'], times.numpy()), \
            f"Time steps in {self.path} differ. Delete the old file or change the grid for time propagation."
        assert previous['propagator'] == propagator_name, "Data produced with different propagators cannot be added."

    def add_batch(self, autocorrelation, ic_correlation, ntraj):
        """fold the means over ``ntraj`` new trajectories into the stored means"""
        stored = dict(np.load(self.path))
        done = stored['trajectories']
        total = done + ntraj
        stored['autocorrelation'] = (ntraj * autocorrelation + done * stored['autocorrelation']) / total
        stored['ic_correlation'] = (ntraj * ic_correlation + done * stored['ic_correlation']) / total
        stored['trajectories'] = total
        stored.pop('ic_rate', None)          # a rate computed from the old correlation function is stale now
        logger.info(f"<phi(0)|phi(0)>= {stored['autocorrelation'][0]}")
        assert abs(stored['autocorrelation'][0] - 1.0) < 1.0e-3
        np.savez(self.path, **stored)


# ---------------------------------------------------------------------------------------------------------------------
# dynamics
# ---------------------------------------------------------------------------------------------------------------------

def make_propagator(task, Gamma_0, device):
    """frozen Gaussians of the width of the initial wavepacket: Gamma_i = Gamma_t = Gamma_0 (cli.py:305-306, 376-383)"""
    if task.get('propagator', 'HK') == "WM":
        cell = task.get('cell_width', 10000.0)
        return propagators.WaltonManolopoulosPropagator(Gamma_0, Gamma_0, cell, cell, device=device)
    return propagators.HermanKlukPropagator(Gamma_0, Gamma_0, device=device)


def propagate_batch(propagator, setup, dt, nt, times, norm_every=0, flush=None, across_ranks=False, log=True):
    """C_auto(t), k_ic(t) of one batch of trajectories: the device loop leaves the raw per-step sums in a device buffer,
    ``flush`` (None on a single rank) adds the buffers of all ranks -- ONE all-reduce per batch, SURVEY 8e -- and the host
    applies the dynamical phase.  ``norm_every`` > 0 logs the wavefunction norm (the O(n^2) convergence diagnostic of
    cli.py:424-429) at every norm_every-th step by cutting the device loop there; with ``across_ranks`` it is the norm of
    the whole sharded batch (a collective: every rank calls it, ``log`` says who prints)."""
    slots = torch.zeros((nt, 5), dtype=torch.float64, device=propagator.device)
    length = norm_every if norm_every > 0 else nt
    pieces = []
    for first in range(0, nt, length):
        if norm_every > 0:
            norm = propagator.norm(across_ranks=across_ranks)
            if log:
                logger.info(f" time/fs= {times[first] * units.autime_to_fs}  norm= {norm:9.6f}")
        count = min(length, nt - first)
        pieces.append((first, count, propagator.t))
        propagator.run(setup.potential, dt, count, slots=slots[first:first + count])
    if flush is not None:
        flush(slots)
    propagator.synchronize()                      # raises the energy-conservation error of this rank's trajectories
    parts = [propagator.finalize_slots(slots[first:first + count], t0, dt, setup.zero_point_energy)
             for first, count, t0 in pieces]
    return tuple(np.concatenate(part) for part in zip(*parts))


def run_semiclassical_dynamics(task, device='cuda', comm=None):
    """One 'dynamics' task.  Under ``torch.distributed`` with more than one rank (``python -m semiclassical_amd.driver
    dynamics input.json --gpus N`` or torchrun; north_star: "trajectory batches shard embarrassingly across the 8 GPUs of
    one node with a single RCCL all-reduce ... per flush") every rank integrates ITS share of each batch of the
    reference's repetition loop (cli.py:321-324, 374-476) on its own GPU with the batch size as Monte-Carlo weight, one
    all-reduce per batch adds the raw sums, and rank 0 keeps the result file.  ``comm``: an
    ``distributed.RcclCommunicator`` to flush through the C-ABI's sc_flush_allreduce instead of the process group."""
    from . import distributed as Dm
    torch.set_default_dtype(torch.float64)
    rank, world = Dm.get_rank(), Dm.world_size()
    if comm is not None:
        rank, world = comm.rank, comm.world
    writer = rank == 0
    setup = build_problem(task)

    dt = task['time_step_fs'] / units.autime_to_fs
    nt = task['num_steps']
    # the stored grid has spacing nt dt / (nt - 1) while the propagator advances by dt (cli.py:312-313)
    times = torch.linspace(0.0, nt * dt, nt)

    batch_size = task.get('batch_size', 10000)
    num_trajectories = task.get('num_trajectories', 50000)
    repetitions = max(num_trajectories // batch_size, 1)
    per_batch = min(batch_size, num_trajectories)
    mine = Dm.shard_slice(per_batch, rank, world)            # this rank's trajectories of every batch
    if mine.stop == mine.start:
        raise ConfigurationError(f"batches of {per_batch} trajectories cannot be shared by {world} ranks")

    store = CorrelationStore(task['results'].get('correlations', 'correlations.npz'))
    if writer:
        store.start(task, task.get('propagator', 'HK'), times, setup)

    seed = task.get('manual_seed', None)
    if seed is not None:
        logger.warning("The random number generator should not be seeded manually unless for debugging!")
    # Where the phase-space points are drawn (a key the reference does not have; it samples on its compute device,
    # cli.py:392 -> propagators.py:537-539): "device" = sc_sample_initial (counter-based Philox keyed by the seed, one
    # subsequence per repetition, nothing crosses PCIe; a rank draws ITS slice of the batch), "host" = torch's CPU
    # generator as in the reference's CPU runs (ranks sharing a batch all draw the whole batch from the same seed).
    sampling = task.get('sampling', 'device')
    if sampling not in ('device', 'host'):
        raise ValueError("'sampling' should be one of 'device' or 'host'")
    if seed is None and world > 1:
        seed_all = Dm.broadcast_object(int.from_bytes(os.urandom(7), 'little'))     # fresh entropy, the same on every rank
    else:
        seed_all = seed
    if seed_all is not None and (seed is not None or sampling == 'host'):
        torch.manual_seed(seed_all)
    device_seed = int(seed_all) if seed_all is not None else int.from_bytes(os.urandom(8), 'little')

    if comm is not None:
        flush = lambda slots: Dm.flush_correlations(slots, comm=comm)
    elif world > 1:
        flush = Dm.flush_correlations
    else:
        flush = None

    for repetition in range(repetitions):
        logger.info(f"*** Repetition {repetition + 1} ***")
        propagator = make_propagator(task, setup.Gamma_0, device)
        count = mine.stop - mine.start
        if sampling == 'device':
            propagator.initial_conditions(setup.q0, setup.p0, setup.Gamma_0, ntraj=count, ntraj_total=per_batch,
                                          seed=device_seed, subsequence=repetition, first_index=mine.start)
        elif world == 1:
            propagator.initial_conditions(setup.q0, setup.p0, setup.Gamma_0, ntraj=per_batch)
        else:
            zi, probi = propagator.draw_initial_conditions(setup.q0, setup.p0, setup.Gamma_0, per_batch)
            propagator.set_initial_conditions(setup.q0, setup.p0, setup.Gamma_0, zi[:, mine], probi[mine],
                                              ntraj_total=per_batch)
        autocorrelation, ic_correlation = propagate_batch(propagator, setup, dt, nt, times,
                                                          norm_every=task.get('calc_norm_every', 0), flush=flush,
                                                          across_ranks=world > 1 and comm is None, log=writer)
        assert not np.isnan(autocorrelation).any(), f"encountered NaN's in autocorrelation : {autocorrelation}"
        assert not np.isnan(ic_correlation).any(), f"encountered NaN's in IC correlation : {ic_correlation}"
        if writer:
            store.add_batch(autocorrelation, ic_correlation, per_batch)


# ---------------------------------------------------------------------------------------------------------------------
# rates
# ---------------------------------------------------------------------------------------------------------------------

def lineshape_from_task(task):
    """damping function of the 'broadening' keys; widths are half widths at half maximum in eV (cli.py:524-545)"""
    kind = task.get('broadening', 'gaussian')
    hwhm_gauss, hwhm_lorentz = task.get('hwhmG_ev', 0.01), task.get('hwhmL_ev', 1.0e-6)
    sigma = hwhm_gauss / np.sqrt(2.0 * np.log(2.0)) / units.hartree_to_ev
    gamma = hwhm_lorentz / units.hartree_to_ev
    shapes = {"gaussian": lambda: broadening.gaussian(sigma),
              "lorentzian": lambda: broadening.lorentzian(gamma),
              "voigtian": lambda: broadening.voigtian(sigma, gamma)}
    if kind not in shapes:
        raise ValueError("'broadening' should be one of 'gaussian', 'lorentzian' or 'voigtian'")
    return shapes[kind](), {'broadening': kind, 'hwhmG': hwhm_gauss, 'hwhmL': hwhm_lorentz}


def calculate_rates(task):
    """k_ic(E) from the stored k_ic(t): damped Fourier transform, factor 2 pi, non-negative energies (cli.py:547-570)"""
    lineshape, record = lineshape_from_task(task)
    stored = dict(np.load(task.get('correlations', 'correlations.npz')))
    stored.update(record)
    energies, spectrum = rates.rate_from_correlation(stored['times'], stored['ic_correlation'], lineshape)
    keep = energies >= 0.0
    stored['energies'] = energies[keep]
    stored['ic_rate'] = (2.0 * np.pi * spectrum)[keep].real
    np.savez(task.get('rates', 'correlations.npz'), **stored)


def main(argv=None):
    logging.basicConfig(format="[%(module)-12s] %(message)s", level=logging.INFO)
    parser = argparse.ArgumentParser(prog="semi (semiclassical_amd)")
    sub = parser.add_subparsers(dest='command')
    dyn = sub.add_parser('dynamics', help="run semiclassical dynamics")
    dyn.add_argument('json_input', type=str, metavar='input.json')
    dyn.add_argument('--cuda', type=int, default=None, metavar='id',
                     help="GPU of this process (default: 0, or LOCAL_RANK in a multi-rank job)")
    dyn.add_argument('--gpus', type=int, default=1, metavar='N',
                     help="share every batch of trajectories among N GPUs of this node: starts N rank processes (one per "
                          "GPU, RCCL process group) unless a launcher (torchrun) has already done so")
    rat = sub.add_parser('rates', help="Fourier transform correlation functions into rates")
    rat.add_argument('json_input', type=str, metavar='input.json')
    args = parser.parse_args(argv)
    if args.command == 'dynamics' and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # self-launch: this parent never touches the GPU, it starts one rank process per GPU and waits for them
        import sys
        from . import distributed as Dm
        forwarded = list(sys.argv[1:] if argv is None else argv)
        raise SystemExit(Dm.launch_local_ranks(["-m", "semiclassical_amd.driver"] + forwarded, args.gpus))
    with open(args.json_input) as f:
        config = json.load(f)
    if args.command == 'dynamics':
        from . import distributed as Dm
        rank, world, local = Dm.init_from_env()
        cuda = args.cuda if args.cuda is not None else (local if world > 1 else 0)
        if world > 1 and rank != 0:
            logging.getLogger().setLevel(logging.WARNING)          # one voice per job
        handler = lambda task: run_semiclassical_dynamics(task, device=f"cuda:{cuda}")
    else:
        world, handler = 1, calculate_rates
    for task in config['semi']:
        if task['task'] == args.command:
            handler(task)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
