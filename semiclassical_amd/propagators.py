"""Semiclassical propagators on MI355X.

Host-side mirror of the reference's propagator interface
(semiclassical/propagators.py: ``HermanKlukPropagator`` :407, same method
names, argument meaning and error behaviour) over the HIP kernels of
``libsemiclassical_hip.so``.  Python only prepares O(D^2) constants and issues
kernel launches; all per-trajectory arithmetic runs on the GPU.

Differences a caller can observe:
  * the state lives in an engine-native, trajectory-major layout; the
    reference's ``(rows, n)`` tensor is produced on demand by the ``y``
    property (and accepted by its setter);
  * ``step()`` does not synchronise: the energy-conservation guard
    (reference propagators.py:385-398) is evaluated on the device and raised,
    with the reference's message, at the next host synchronisation point
    (``autocorrelation()``, ``ic_correlation()``, ``synchronize()``, ``run()``);
  * ``run()`` executes the caller loop of cli.py:401-436 (correlate,
    correlate, step -- nt times) without any host synchronisation.
"""
import logging

import numpy as np
import torch

from . import _lib, hostmath
from ._lib import (lib, check, ptr, sc_state, sc_hk_consts, sc_overlap_consts, sc_nac_consts, sc_wm_consts,
                   sc_dense_scratch, sc_multi_scratch)
from .units import hbar

__all__ = ['HermanKlukPropagator', 'WaltonManolopoulosPropagator']

logger = logging.getLogger(__name__)

C128 = torch.complex128
F64 = torch.float64


def _potential_fingerprint(potential):
    """Identity of the constants a potential contributes to the cached device buffers (1/m, coupling vectors).

    Caches are keyed on the object itself (a strong reference, compared with ``is``: a recycled ``id`` cannot alias a
    dead potential) AND on this cheap content fingerprint, so that masses or coupling vectors changed in place are
    picked up at the next step / run entry.  O(D) host work per call.

    The derivative couplings are probed at TWO points.  The reference evaluates them at the initial and the current
    position of every trajectory (propagators.py:880-892, 1685-1693); all of its own potentials return constant vectors,
    which the kernels take as constants.  A potential whose couplings differ between the probes is position dependent:
    the last entry of the fingerprint says so and the propagators evaluate its couplings per trajectory.
    """
    d = potential.dimensions()
    probe = torch.zeros((d, 1), dtype=F64)
    other = (0.05 + 0.1 * torch.arange(1, d + 1, dtype=F64) / d).reshape(d, 1)
    masses = hostmath.as_f64(potential.masses())
    tau1 = hostmath.as_f64(potential.derivative_coupling_1st(probe))[:, 0]
    tau2 = hostmath.as_f64(potential.derivative_coupling_2nd(probe))[:, 0]
    tau1b = hostmath.as_f64(potential.derivative_coupling_1st(other))[:, 0]
    tau2b = hostmath.as_f64(potential.derivative_coupling_2nd(other))[:, 0]
    varies = not (torch.equal(tau1, tau1b) and torch.equal(tau2, tau2b))
    return masses, tau1, tau2, torch.tensor([float(varies)], dtype=F64)


def _same_fingerprint(a, b):
    return a is not None and b is not None and all(x.shape == y.shape and torch.equal(x, y) for x, y in zip(a, b))


def _resolve_device(device):
    dev = torch.device(device)
    if dev.type != 'cuda':
        raise RuntimeError(
            f"semiclassical_amd propagators run on an AMD GPU (device='cuda[:i]'), got device='{device}'. "
            "There is no CPU path.")
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    return dev


class HermanKlukPropagator(object):
    """Herman-Kluk frozen-Gaussian propagator (reference propagators.py:407-1066)."""

    def __init__(self, Gamma_i, Gamma_t, device='cuda', exploit_separability=False):
        """``exploit_separability`` (not in the reference): opt in to the O(D) diagonal-state kernel whenever the
        potential is separable, the width matrices are diagonal and the monodromy matrices are still the diagonal
        ones ``initial_conditions`` created -- the dense blocks the reference carries are then never formed on the
        device unless ``y`` / ``monodromy_matrices()`` ask for them ("separable shortcut")."""
        self.exploit_separability = bool(exploit_separability)
        Gamma_i, Gamma_t = hostmath.as_f64(Gamma_i), hostmath.as_f64(Gamma_t)
        assert hostmath.is_symmetric_non_negative(Gamma_i), "Gamma_i has to be symmetric and positive semi-definite."
        assert hostmath.is_symmetric_non_negative(Gamma_t), "Gamma_t has to be symmetric and positive semi-definite."
        self.device = _resolve_device(device)
        self._Gi, self._Gt = Gamma_i, Gamma_t                 # host copies (setup algebra)
        self.Gamma_i, self.Gamma_t = Gamma_i.to(self.device), Gamma_t.to(self.device)
        self.sqGi, self.isqGi = hostmath.sym_sqrtm(Gamma_i)
        self.sqGt, self.isqGt = hostmath.sym_sqrtm(Gamma_t)
        self._nac, self._nac_pot, self._nac_fp, self._nac_generic = None, None, None, None
        self._ntraj_norm = None

    # ------------------------------------------------------------------ initial conditions
    def initial_conditions(self, q0, p0, Gamma_0, ntraj=5000, ntraj_total=None, generator=None, seed=None, subsequence=0,
                           first_index=0):
        """Sample ``ntraj`` phase-space points from |<qi,pi,Gamma_i|q0,p0,Gamma_0>|^2 and reset the state.

        Two sources of the standard-normal deviates xi (reference propagators.py:537-539):

        * default / ``generator``: drawn on the HOST with the call the reference makes, so on a given torch build
          ``torch.manual_seed(s)`` reproduces the reference's CPU initial conditions (parity fixtures are made this way);
        * ``seed`` (an integer): drawn ON THE DEVICE by ``sc_sample_initial`` (Philox4x32-10 + Box-Muller), together with
          ``zi = z0 + iLz^T xi``, ``probi`` and the state of t = 0 -- nothing crosses PCIe.  Deviate j of the trajectory
          with global index ``first_index + i`` depends only on ``(seed, subsequence, first_index + i, j)``: a rank that
          owns one shard of a larger batch passes its first global index (same ensemble as one big batch) or its own
          ``subsequence`` (an independent ensemble).

        ``ntraj_total`` (default ``ntraj``) is the N of the Monte-Carlo weight 1/(N P(qi,pi)); a rank that owns one shard
        of a larger batch passes the global count.
        """
        q0, p0, Gamma_0 = hostmath.as_f64(q0), hostmath.as_f64(p0), hostmath.as_f64(Gamma_0)
        assert Gamma_0.size() == self._Gi.size(), "Width parameter matrix Gamma_0 has wrong dimensions."
        assert hostmath.is_symmetric_non_negative(Gamma_0), "Gamma_0 has to be symmetric and positive semi-definite."
        d = q0.size()[0]
        U, iGi0, iLz, detLz, dprime = hostmath.sampling_matrices(self._Gi, Gamma_0)
        z0 = torch.cat((q0, p0))
        prob0 = detLz / (2 * np.pi) ** d
        if seed is not None:
            assert generator is None, "either a host generator or a device seed"
            if not 0 <= int(subsequence) < 2 ** 56:
                raise ValueError(f"subsequence {subsequence} outside [0, 2^56)")
            dev = self.device
            self._begin_state(q0, p0, Gamma_0, U, iGi0, int(ntraj), ntraj_total)
            self._zi_t = torch.empty((ntraj, 2 * d), dtype=F64, device=dev)
            self.probi = torch.empty(ntraj, dtype=F64, device=dev)
            ilz_d, z0_d = iLz.contiguous().to(dev), z0.to(dev)
            check(lib.sc_sample_initial(self._state, ptr(ilz_d), ptr(z0_d), dprime, float(prob0), int(seed) & (2 ** 64 - 1),
                                        int(subsequence), int(first_index), 1, ptr(self._zi_t),
                                        ptr(self.probi), None, self._stream()))
            self.zi = self._zi_t.t()                   # the reference's (2D, n) attribute, as a view
            self._finish_state((ilz_d, z0_d))
            return
        zi, probi = self._draw_on_host(z0, iLz, prob0, dprime, ntraj, generator)
        self.set_initial_conditions(q0, p0, Gamma_0, zi, probi, ntraj_total=ntraj_total)

    @staticmethod
    def _draw_on_host(z0, iLz, prob0, dprime, ntraj, generator=None):
        if generator is None:
            xi = torch.distributions.Normal(torch.zeros(2 * dprime, dtype=F64),
                                            torch.ones(2 * dprime, dtype=F64)).sample((ntraj,)).T
        else:
            xi = torch.randn((ntraj, 2 * dprime), dtype=F64, generator=generator).T
        zi = z0.unsqueeze(1) + torch.einsum('ji,jn->in', iLz, xi)
        probi = prob0 * torch.exp(-0.5 * torch.einsum('in,in->n', xi, xi))
        return zi, probi

    def draw_initial_conditions(self, q0, p0, Gamma_0, ntraj, generator=None):
        """The host draw of ``initial_conditions`` without touching the state: ``(zi (2D, ntraj), probi (ntraj,))`` from
        torch's CPU generator with the call the reference makes (propagators.py:537-566).  Ranks that share one batch
        all draw the whole batch from the same seed and hand their slice to ``set_initial_conditions``."""
        q0, p0, Gamma_0 = hostmath.as_f64(q0), hostmath.as_f64(p0), hostmath.as_f64(Gamma_0)
        _, _, iLz, detLz, dprime = hostmath.sampling_matrices(self._Gi, Gamma_0)
        prob0 = detLz / (2 * np.pi) ** q0.size()[0]
        return self._draw_on_host(torch.cat((q0, p0)), iLz, prob0, dprime, int(ntraj), generator)

    def set_initial_conditions(self, q0, p0, Gamma_0, zi, probi, ntraj_total=None):
        """Start from given phase-space points ``zi`` (2D, n) with sampling densities ``probi`` (n,)."""
        q0, p0, Gamma_0 = hostmath.as_f64(q0), hostmath.as_f64(p0), hostmath.as_f64(Gamma_0)
        dev = self.device
        d, n = q0.size()[0], zi.shape[1]
        assert zi.shape[0] == 2 * d and probi.shape[0] == n
        U, iGi0, _, _, dprime = hostmath.sampling_matrices(self._Gi, Gamma_0)
        self._begin_state(q0, p0, Gamma_0, U, iGi0, n, ntraj_total)
        self.zi = torch.as_tensor(zi, dtype=F64).to(dev).contiguous()
        self.probi = torch.as_tensor(probi, dtype=F64).to(dev).contiguous()
        self._zi_t = self.zi.t().contiguous()                          # [n][2D]
        self._qp.copy_(self._zi_t)
        self._act.zero_()
        self._mono.zero_()
        torch.diagonal(self._mono[:, 0], dim1=1, dim2=2).fill_(1.0)
        torch.diagonal(self._mono[:, 3], dim1=1, dim2=2).fill_(1.0)
        self._c2.fill_(1.0)
        self._sgn.fill_(1.0)
        self._finish_state(None)

    def _begin_state(self, q0, p0, Gamma_0, U, iGi0, n, ntraj_total):
        """host constants + the (uninitialised) engine state of a batch of ``n`` trajectories"""
        dev = self.device
        d = q0.size()[0]
        self.dim, self.ntraj = d, n
        self._ntraj_norm = n if ntraj_total is None else int(ntraj_total)
        self._q0h, self._p0h, self._G0h, self._iGi0h = q0, p0, Gamma_0, iGi0
        self.q0, self.p0, self.Gamma_0 = q0.to(dev), p0.to(dev), Gamma_0.to(dev)
        self.U, self.iGi0 = U.to(dev), iGi0.to(dev)
        # ---- engine state (trajectory-major) ----
        self._qp = torch.empty((n, 2 * d), dtype=F64, device=dev)
        self._act = torch.empty(n, dtype=F64, device=dev)
        self._mono = torch.empty((n, 4, d, d), dtype=F64, device=dev)
        self._c2 = torch.empty(n, dtype=C128, device=dev)
        self._sgn = torch.empty(n, dtype=F64, device=dev)
        self._flags = torch.zeros(n + 2, dtype=torch.int32, device=dev)     # [n] = flagged count, [n + 1] = work cursor
        self._work = torch.zeros((n, 4, d), dtype=F64, device=dev)
        self._state = sc_state(n=n, dim=d, mono_layout=_lib.SC_MONO_ROWMAJOR, qp=ptr(self._qp), act=ptr(self._act),
                               mono=ptr(self._mono), c2=ptr(self._c2), sgn=ptr(self._sgn), work=ptr(self._work),
                               flags=ptr(self._flags))

    def _finish_state(self, keep_alive):
        """per-step scratch, constants and the prefactor of t = 0 once (q, p, S, M) and (zi, probi) are in place"""
        dev, n, d = self.device, self.ntraj, self.dim
        self._ic_bufs = keep_alive                    # device operands of a sampling launch that may still be running
        self._gstep = lib.sc_step_grid(n, d)
        self._gcorr = lib.sc_correlate_grid(n, d)
        self._epart = torch.zeros(self._gstep, dtype=F64, device=dev)
        self._cpart = torch.zeros((self._gcorr, 4), dtype=F64, device=dev)
        self._cq = torch.zeros(n, dtype=C128, device=dev)
        self._kq = torch.zeros(n, dtype=C128, device=dev)
        self._slot = torch.zeros(8, dtype=F64, device=dev)
        self._elog = torch.zeros(4, dtype=F64, device=dev)            # energy guard log (sc_energy_guard)
        self._nsteps = 0
        self._corr_step, self._corr_has_nac = -1, False
        self._nac, self._nac_pot, self._nac_fp, self._nac_generic = None, None, None, None
        self._dense = None
        # diagonals of the monodromy blocks for the separable shortcut; _mono is stale while they are ahead of it
        self._mdiag = torch.zeros((n, 4, d), dtype=F64, device=dev)
        self._mdiag[:, 0] = 1.0
        self._mdiag[:, 3] = 1.0
        self._mono_is_diag, self._mono_stale = True, False
        # The monodromy blocks are the identity now and stay DIAGONAL as long as only separable potentials act on them.  Nothing in
        # the dense-state kernels uses that -- except that diagonal blocks cannot meet a weak pivot, which is what lets run() take
        # two time steps per visit (sc_hk_step_multi: an intermediate determinant cannot be repaired after the fact).
        self._blocks_structurally_diagonal = True
        self._multi = None

        self._prepare()
        self.t = 0.0
        self._prefactor_initial()

    def _prepare(self):
        """constants of the three overlaps and of the prefactor (reference propagators.py:633-643)"""
        dev = self.device
        self._pre = hostmath.PrefactorConstants(self._Gi, self._Gt, self.U.cpu())
        up = lambda x: x.contiguous().to(dev)
        self._pre_bufs = ([up(self._pre.st), up(self._pre.si)] if self._pre.diag else
                          [up(self._pre.L1), up(self._pre.L2), up(self._pre.R1), up(self._pre.R2)])
        b = self._pre_bufs
        if self._pre.diag:
            self._hk = sc_hk_consts(dim=self.dim, dprime=self.dim, diag=1, st=ptr(b[0]), si=ptr(b[1]))
        else:
            # imaginary parts of U^T Gt^(+-1/2), Gi^(-+1/2) U: zero for positive semi-definite widths up to rounding dust
            # (1e-24 for methylium); "real" = below one ulp of the largest entry, so dropping them changes no bit
            real = all(float(m.imag.abs().max()) <= 1e-17 * float(m.real.abs().max()) for m in
                       (self._pre.L1, self._pre.L2, self._pre.R1, self._pre.R2))
            self._hk = sc_hk_consts(dim=self.dim, dprime=self._pre.dprime, diag=0, real_lr=int(real),
                                    L1=ptr(b[0]), L2=ptr(b[1]), R1=ptr(b[2]), R2=ptr(b[3]))
        self._ovl_i0, self._ovl_i0_bufs = self._overlap_consts(self._Gi, self._G0h)
        self._ovl_t0, self._ovl_t0_bufs = self._overlap_consts(self._Gt, self._G0h)
        # <qi,pi,Gamma_i|phi(0)> does not depend on time: evaluate once (reference recomputes it every step, :794-795)
        self._vi = torch.zeros(self.ntraj, dtype=C128, device=dev)
        check(lib.sc_overlap(self._ovl_i0, ptr(self._zi_t), self.ntraj, ptr(self._vi), self._stream()))

    def _overlap_consts(self, Gbra, Gket):
        oc = hostmath.OverlapConstants(Gbra, Gket)
        dev = self.device
        if oc.diag:
            mats = [torch.diagonal(m).contiguous().to(dev) for m in (oc.A, oc.B, oc.C)]
        else:
            mats = [m.contiguous().to(dev) for m in (oc.A, oc.B, oc.C)]
        bufs = mats + [self.q0, self.p0]
        c = sc_overlap_consts(dim=self.dim, diag=int(oc.diag), A=ptr(bufs[0]), B=ptr(bufs[1]), C=ptr(bufs[2]),
                              qk=ptr(self.q0), pk=ptr(self.p0), fac=oc.fac)
        return c, bufs

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _prefactor_initial(self):
        """prefactor at t = 0 and initialisation of the branch tracker (reference propagators.py:631)"""
        if self.dim > 64:
            check(lib.sc_dense_mono_step(self._state, self._hk, None, None, self._mono_sums_ptr(), 0.0, 1, self._stream()))
        else:
            check(lib.sc_hk_step(_NULL_POT(self.dim), self._state, self._hk, 0.0, 1, None, self._stream()))
        self._after_prefactor(track=2)

    # ------------------------------------------------------------------ time stepping
    def step(self, potential, dt):
        """propagate all trajectories by one RK4 step t -> t+dt (reference propagators.py:645-655)"""
        assert self.dim == potential.dimensions(), "potential has wrong dimensions"
        self._launch_step(potential, float(dt))
        self.t += float(dt)

    # ---- optional per-kernel timing (bench.py): HIP events on the launch stream around labelled launches ----
    kernel_timing = False

    def _timed(self, label):
        return _KernelTimer(self, label) if self.kernel_timing else _NO_TIMER

    def kernel_times_ms(self):
        """{label: [duration of every launch recorded while ``kernel_timing`` was set]}; synchronises"""
        torch.cuda.current_stream(self.device).synchronize()
        return {k: [a.elapsed_time(b) for a, b in v] for k, v in self.__dict__.get("_kernel_events", {}).items()}

    def _launch_step(self, potential, dt, desc=None, remembered=False):
        """``desc`` / ``remembered``: run() resolves the potential's device descriptor and coupling constants once per
        call instead of once per step"""
        s = self._stream()
        timed = getattr(self, "profile_step_kernel", False)
        if timed:
            # HIP events on the launch stream, bracketing only the step kernel(s) (bench.py roofline)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if hasattr(potential, "_gdml_model"):
            self._blocks_structurally_diagonal = False
            self._sync_dense_mono(leave_diagonal=True)
            self._set_mono_layout(_lib.SC_MONO_ROWMAJOR)
            nblocks = self._launch_dense_step(potential, dt, s)
        elif not hasattr(potential, "_descriptor") or self.dim > 64:
            # no device descriptor, or beyond the fused kernels' D <= 64: the potential's own torch code + dense path
            self._blocks_structurally_diagonal = False
            self._sync_dense_mono(leave_diagonal=True)
            self._set_mono_layout(_lib.SC_MONO_ROWMAJOR)
            nblocks = self._launch_generic_step(potential, dt, s)
        else:
            if desc is None:
                desc = self._potential_descriptor(potential, dt)
            if desc.kind not in (_lib.SC_POT_MORSE, _lib.SC_POT_HARMONIC_SEP, _lib.SC_POT_EPS_MORSE):
                self._blocks_structurally_diagonal = False          # a dense Hessian couples the rows
            if self._shortcut_applies(desc):
                check(lib.sc_hk_step_diag(desc, self._state, self._hk, ptr(self._mdiag), dt, 0, ptr(self._epart), s))
                self._mono_stale = True
            else:
                self._sync_dense_mono(leave_diagonal=True)
                self._set_mono_layout(self._fast_path_layout(desc))
                with self._timed("hk_step"):
                    check(lib.sc_hk_step(desc, self._state, self._hk, dt, 0, ptr(self._epart), s))
            nblocks = self._gstep
        if timed:
            e1.record()
            self.__dict__.setdefault("_step_events", []).append((e0, e1, 1))
        check(lib.sc_energy_guard(ptr(self._epart), nblocks, float(self.ntraj), ptr(self._elog), s))
        self._nsteps += 1
        if not remembered:
            self._remember_nac(potential)
        self._after_prefactor(track=1)

    def _shortcut_applies(self, desc):
        return (self.exploit_separability and self._mono_is_diag and bool(self._pre.diag)
                and desc.kind in (_lib.SC_POT_MORSE, _lib.SC_POT_HARMONIC_SEP, _lib.SC_POT_EPS_MORSE))

    # The separable fast path streams the monodromy blocks in 16 x 16 tiles (SC_MONO_TILED16, include/semiclassical_hip.h);
    # everything else reads them row-major.  The state switches order in place when it enters / leaves that path.
    _tiled_fast_path = True

    def _fast_path_layout(self, desc):
        fast = (self._tiled_fast_path and bool(self._pre.diag) and 16 < self.dim <= 64
                and desc.kind in (_lib.SC_POT_MORSE, _lib.SC_POT_HARMONIC_SEP, _lib.SC_POT_EPS_MORSE))
        return _lib.SC_MONO_TILED16 if fast else _lib.SC_MONO_ROWMAJOR

    def _set_mono_layout(self, layout):
        if self._state.mono_layout != layout:
            check(lib.sc_mono_convert(self._state, layout, self._stream()))
            self._state.mono_layout = layout

    def _sync_dense_mono(self, leave_diagonal=False):
        """bring the dense monodromy blocks up to date with the diagonals the shortcut kernel advanced"""
        if self._mono_stale:
            self._state.mono_layout = _lib.SC_MONO_ROWMAJOR       # rebuilt from the diagonals below
            self._mono.zero_()
            torch.diagonal(self._mono, dim1=2, dim2=3).copy_(self._mdiag)
            self._mono_stale = False
        if leave_diagonal:
            self._mono_is_diag = False          # a dense kernel takes over: the diagonals are no longer maintained

    def _launch_dense_step(self, potential, dt, s):
        """unfused RK4 step for a dense, position-dependent Hessian: four stage kernels + the monodromy kernel"""
        self._dense_scratch()
        with torch.cuda.device(self.device):
            model = potential._gdml_model(self.device)
        for stage in range(4):
            with self._timed("gdml_stage"):
                check(lib.sc_gdml_stage(model, self._state, self._dense, dt, stage, ptr(self._epart), s))
        with self._timed("dense_mono_step"):
            check(lib.sc_dense_mono_step(self._state, self._hk, model.inv_mass, ptr(self._dense_bufs[0]),
                                         self._mono_sums_ptr(), dt, 0, s))
        return self._gdense

    def _dense_scratch(self):
        """scratch of the unfused dense path: stage Hessians, slopes of the previous stage, weighted slope sums"""
        n, d = self.ntraj, self.dim
        if getattr(self, "_dense", None) is None:
            dev = self.device
            bufs = [torch.empty((n, 4, d, d), dtype=F64, device=dev), torch.zeros((n, 2 * d), dtype=F64, device=dev),
                    torch.zeros((n, 2 * d), dtype=F64, device=dev), torch.zeros(n, dtype=F64, device=dev)]
            self._dense_bufs = bufs
            self._dense = sc_dense_scratch(hess=ptr(bufs[0]), kprev=ptr(bufs[1]), ksum=ptr(bufs[2]), ssum=ptr(bufs[3]))
            self._gdense = lib.sc_dense_grid(n)
            if self._gdense > self._epart.numel():
                self._epart = torch.zeros(self._gdense, dtype=F64, device=dev)
        return self._dense

    def _mono_sums_ptr(self):
        """scratch of sc_dense_mono_step beyond D = 64: the RK4 sums of the MFMA kernel up to D = 96, the per-workgroup
        matrix blocks of the any-dimension kernels beyond (sc_dense_mono_scratch_bytes)"""
        need = lib.sc_dense_mono_scratch_bytes(self.ntraj, self.dim, self._hk.dprime)
        assert need >= 0
        if need == 0:
            return None
        if getattr(self, "_mono_sums", None) is None or self._mono_sums.numel() * 8 < need:
            self._mono_sums = torch.empty((need + 7) // 8, dtype=F64, device=self.device)
        return ptr(self._mono_sums)

    def _launch_generic_step(self, potential, dt, s):
        """Any object with the reference's potential protocol (potentials.py:41-204: ``harmonic_approximation(r) ->
        V (n,), grad (D,n), hess (D,D,n)``, ``masses()``): the potential is evaluated by ITS OWN torch code on the
        device at the four RK4 stage points; the RK4 bookkeeping, the monodromy GEMMs and the prefactor stay in HIP
        (generic Python potentials take the unfused path)."""
        n, d = self.ntraj, self.dim
        dense = self._dense_scratch()
        masses = hostmath.as_f64(potential.masses())
        cached = getattr(self, "_generic_masses", None)
        if getattr(self, "_generic_pot", None) is not potential or cached is None or not torch.equal(cached, masses):
            self._generic_inv_mass = (1.0 / masses).to(self.device).contiguous()
            self._generic_r = torch.empty((n, d), dtype=F64, device=self.device)
            self._generic_pot, self._generic_masses = potential, masses.clone()
        r, inv_mass = self._generic_r, self._generic_inv_mass
        for stage in range(4):
            check(lib.sc_stage_point(self._state, dense, dt, stage, ptr(r), s))
            V, grad, hess = potential.harmonic_approximation(r.t())
            assert V.shape == (n,) and grad.shape == (d, n) and hess.shape == (d, d, n), \
                "harmonic_approximation has to return V (n,), grad (D,n), hess (D,D,n)"
            Vc = V.to(F64).contiguous()
            gt = grad.to(F64).t().contiguous()
            # the monodromy kernel takes the stage Hessian image as A[i][k] = image[k][i]: store the transpose
            self._dense_bufs[0][:, stage].copy_(hess.permute(2, 1, 0))
            check(lib.sc_stage_consume(self._state, dense, ptr(inv_mass), ptr(Vc), ptr(gt), dt, stage,
                                       ptr(self._epart), s))
        check(lib.sc_dense_mono_step(self._state, self._hk, ptr(inv_mass), ptr(self._dense_bufs[0]), self._mono_sums_ptr(),
                                     dt, 0, s))
        return self._gdense

    def _after_prefactor(self, track):
        """hook for propagators whose prefactor needs more than the HK determinant (WM)"""

    def _potential_descriptor(self, potential, dt=None):
        if not hasattr(potential, "_descriptor"):
            raise TypeError(f"{type(potential).__name__} has no device descriptor: it takes the unfused path "
                            "(_launch_generic_step), not a fused kernel")
        with torch.cuda.device(self.device):
            return potential._descriptor(self.device) if dt is None else potential._descriptor(self.device, dt)

    def _check_energy_guard(self):
        """raise the reference's RuntimeError (propagators.py:396) if <T+V> jumped by more than 1e-2 Hartree"""
        change = float(self._elog[2].item())
        if change > 1.0e-2:
            raise RuntimeError(f"average energy of classical trajectories is not conserved, change= {change} Hartree")

    def synchronize(self):
        torch.cuda.current_stream(self.device).synchronize()
        self._run_scratch = None              # partial sums of the last whole-loop run(): consumed by now
        if self._multi is not None and int(self._multi["bad"].item()) != 0:
            raise _lib.EngineError("sc_hk_step_multi met a weak pivot in an intermediate determinant although the monodromy blocks "
                                   "were taken to be diagonal: the results of this run() are not reliable")
        self._check_energy_guard()

    def mean_energy(self):
        """<T+V> over the trajectories as the energy guard saw it at the last step: the value at the k4 stage point
        (reference propagators.py:380 ``_en_mean``, quirk Q2), not that of the accepted state.  Host sync."""
        return float(self._elog[1].item())

    def step_kernel_times_ms(self):
        """duration PER TIME STEP of the step-kernel launches recorded while ``profile_step_kernel`` was set (a launch of the
        two-steps-per-visit path counts for two)"""
        torch.cuda.current_stream(self.device).synchronize()
        return [e0.elapsed_time(e1) / steps for e0, e1, steps in self.__dict__.get("_step_events", [])]

    # ------------------------------------------------------------------ correlation functions
    def _nac_is_current(self, potential):
        """(is the cached coupling data that of `potential`?, fingerprint of `potential`)"""
        fp = _potential_fingerprint(potential)
        return (self._nac_pot is potential and _same_fingerprint(self._nac_fp, fp)), fp

    def _remember_nac(self, potential):
        current, fp = self._nac_is_current(potential)
        if current:
            return
        masses, tau1, tau2, varies = fp
        dev = self.device
        self._nac_pot, self._nac_fp = potential, tuple(x.clone() for x in fp)
        self._corr_step = -1
        if bool(varies.item()):
            # position-dependent couplings: evaluated per trajectory by the potential's own torch code (_nac_terms)
            self._nac, self._nac_bufs, self._nacq = None, None, None
            G = self._G0h @ self._iGi0h
            self._nac_generic = {"minv": (1.0 / masses).to(dev).unsqueeze(1), "R": (G @ self._Gi).to(dev), "G": G.to(dev)}
            self._nac_generic["initial"] = self._nac_terms(potential, self.zi[:self.dim], self.zi[self.dim:], sign=+1.0)
            return
        self._nac_generic = None
        nc = hostmath.NacConstants(self._G0h, self._Gi, self._iGi0h, self._p0h, masses, tau1,
                                   tau2_sum=float(torch.sum(tau2 / masses)))
        bufs = [nc.rn.to(dev), nc.gn.to(dev)]
        self._nac = sc_nac_consts(dim=self.dim, rn=ptr(bufs[0]), gn=ptr(bufs[1]), q0=ptr(self.q0), p0=ptr(self.p0),
                                  p0n1=nc.p0n1, n2=nc.n2)
        self._nac_bufs = bufs
        self._nacq = torch.zeros(self.ntraj, dtype=C128, device=dev)
        check(lib.sc_nac_initial(self._nac, ptr(self._zi_t), self.ntraj, ptr(self._nacq), self._stream()))

    def _coupling_vectors(self, potential, r):
        """n1 (D, n) and n2 (n,) of eqns (89), (90) at the positions r (D, n) on the device, reference propagators.py:880-892"""
        c = self._nac_generic
        tau1 = potential.derivative_coupling_1st(r).to(F64)
        tau2 = potential.derivative_coupling_2nd(r).to(F64)
        assert tau1.shape == r.shape and tau2.shape == r.shape, "derivative couplings have to be of shape (D, n)"
        return -hbar ** 2 * c["minv"] * tau1, -hbar ** 2 * 0.5 * torch.sum(c["minv"] * tau2, dim=0)

    def _nac_terms(self, potential, r, mom, sign):
        """nacq (sign = +1, initial points) or nacQ (sign = -1, current points) of propagators.py:894-903 for
        position-dependent couplings, (n,) complex on the device"""
        c = self._nac_generic
        n1, n2 = self._coupling_vectors(potential, r)
        PI = self.p0.unsqueeze(1) + c["G"] @ (mom - self.p0.unsqueeze(1))
        real = n2 + torch.sum((self.q0.unsqueeze(1) - r) * (c["R"] @ n1), dim=0)
        return torch.complex(real, sign / hbar * torch.sum(PI * n1, dim=0))

    def _generic_kic(self, slot_row):
        """k_ic sum of the current state for position-dependent couplings into slot_row[2:4] (device, no host sync):
        sum_i nacQ_i nacq_i cq_i / hbar^2 with the per-trajectory C_auto terms the correlate kernel just wrote"""
        c = self._nac_generic
        qp = self._qp.t()
        nacQ = self._nac_terms(self._nac_pot, qp[:self.dim], qp[self.dim:], sign=-1.0)
        kq = nacQ * c["initial"] * self._cq / hbar ** 2
        self._kq.copy_(kq)
        total = torch.sum(kq)
        slot_row[2:4] = torch.view_as_real(total)

    def _mc_norm(self):
        return self._ntraj_norm * (2 * np.pi * hbar) ** self.dim

    def _reduce_into(self, partials, count, slot_ptr, cursor):
        """sums of the per-workgroup partials into the slot at `slot_ptr`, or into row *cursor of the slot buffer"""
        if cursor is None:
            check(lib.sc_reduce_slot(ptr(partials), count, None, 0, 1.0, C_void(slot_ptr), self._stream()))
        else:
            check(lib.sc_reduce_slot_at(ptr(partials), count, C_void(slot_ptr), ptr(cursor), self._stream()))

    def _launch_correlate(self, slot_ptr, per_trajectory=True, cursor=None, slot_row=None, state=None):
        """per-trajectory terms + their sums for the current state into the 5-double slot at `slot_ptr` (`slot_row`: the same
        five doubles as a tensor, needed when the k_ic sum is formed by torch: position-dependent couplings; `state`: another
        view of (q, p, S, c2, sign) -- the state between the two steps of sc_hk_step_multi)"""
        s = self._stream()
        nac = self._nac
        generic = getattr(self, "_nac_generic", None) is not None
        per_trajectory = per_trajectory or generic
        with self._timed("hk_correlate"):
            check(lib.sc_hk_correlate(self._state if state is None else state, self._ovl_t0, nac, ptr(self._vi), ptr(self.probi),
                                      ptr(self._nacq) if nac is not None else None, self._mc_norm(),
                                      ptr(self._cq) if per_trajectory else None,
                                      ptr(self._kq) if per_trajectory else None, ptr(self._cpart), s))
        self._reduce_into(self._cpart, self._gcorr, slot_ptr, cursor)
        if generic:
            assert slot_row is not None and cursor is None
            self._generic_kic(slot_row)

    def _correlate_current(self, need_nac):
        if self._corr_step == self._nsteps and (self._corr_has_nac or not need_nac):
            return
        self._launch_correlate(self._slot.data_ptr(), slot_row=self._slot)
        self._corr_step, self._corr_has_nac = self._nsteps, (self._nac is not None or self._nac_generic is not None)
        self._slot_host = self._slot.cpu().numpy().copy()          # host sync
        self._check_energy_guard()

    def autocorrelation_qp(self):
        """per-trajectory terms of the autocorrelation function (reference propagators.py:784-807)"""
        self._correlate_current(False)
        return self._cq * (self._mc_norm() * self.probi)

    def autocorrelation(self, energy0_es=0.0):
        """C_auto(t) = e^{i t E0/hbar} <phi(0)|phi(t)> for the current step (reference propagators.py:809-843)"""
        self._correlate_current(False)
        c = complex(self._slot_host[0], self._slot_host[1])
        return c * np.exp(1j / hbar * self.t * energy0_es)

    def ic_correlation(self, potential, energy0_es=0.0):
        """k_ic(t) for the current step (reference propagators.py:845-911)"""
        self._remember_nac(potential)
        self._correlate_current(True)
        k = complex(self._slot_host[2], self._slot_host[3])
        return k * np.exp(1j / hbar * self.t * energy0_es)

    def run(self, potential, dt, nt, energy0_es=0.0, slots=None, use_graph=False):
        """The caller loop of cli.py:401-436 on the device: ``nt`` times (C_auto, k_ic, step), no host sync.

        Returns ``(autocorrelation[nt], ic_correlation[nt])`` as complex NumPy arrays.  With ``slots`` (a
        device tensor (nt, 5)) the raw sums are left on the device for a later flush (see distributed.py)
        and ``None`` is returned.  Columns 0..3 of row k are Re C, Im C, Re k, Im k of the state before step k.
        Column 4 belongs to the engine: the whole-loop kernel (``sc_hk_run``) leaves the mean <T+V> of step k there,
        the step-at-a-time paths do not touch it -- do not keep data of your own in it across ``run()``.

        ``use_graph``: for small batches the loop is bound by the ~6 kernel launches per step, not by the kernels.
        The first iteration then runs as usual and the launch sequence of the second one is captured in a HIP graph
        that is replayed for the remaining steps (the row of ``slots`` a step writes is a device-resident cursor, so
        no kernel argument changes between steps).  Only for the fused kernels (device potential descriptor, D <= 64).

        Separable potentials with diagonal width matrices and 16 < D <= 64 advance TWO time steps per launch while the monodromy
        blocks are known to be diagonal (``sc_hk_step_multi``: the second step's reads come from the memory-side cache; results
        bit-identical to one launch per step; ``pair_steps = False`` switches it off).
        """
        assert self.dim == potential.dimensions(), "potential has wrong dimensions"
        dt = float(dt)
        self._remember_nac(potential)
        own = slots is None
        if own:
            slots = torch.zeros((nt, 5), dtype=F64, device=self.device)
        else:
            self._check_slots(slots, nt)
        t0 = self.t
        base = slots.data_ptr()
        fused = (hasattr(potential, "_descriptor") and not hasattr(potential, "_gdml_model") and self.dim <= 64
                 and self._nac_generic is None)
        desc = self._potential_descriptor(potential, dt) if fused else None
        if fused and self._whole_loop_applies(desc):
            # separable potential, diagonal widths, D <= 12: the whole loop as ONE launch (sc_hk_run)
            self._run_whole_loop(desc, dt, nt, slots, potential)
        elif use_graph and fused and nt > 2 and not getattr(self, "profile_step_kernel", False) and not self.kernel_timing:
            self._run_graph(potential, dt, nt, desc, slots)
        else:
            pairs = fused and nt >= 2 and self._multi_applies(desc)
            k = 0
            while k < nt:
                self._launch_correlate(base + 40 * k, per_trajectory=False, slot_row=slots[k])
                if pairs and k + 1 < nt:
                    # TWO time steps per visit of a trajectory (sc_hk_step_multi): the second step's loads of the monodromy blocks
                    # hit the memory-side cache instead of HBM; its correlation terms come from the state between the two steps
                    self._launch_step_pair(desc, dt)
                    self._launch_correlate(base + 40 * (k + 1), per_trajectory=False, state=self._multi["state_mid"])
                    self.t += dt
                    self.t += dt
                    k += 2
                else:
                    self._launch_step(potential, dt, desc=desc, remembered=True)
                    self.t += dt
                    k += 1
        self._corr_step = -1
        if not own:
            return None
        self.synchronize()
        return self.finalize_slots(slots, t0, dt, energy0_es)

    _whole_loop_ok = True           # WM needs its own per-step kernel between the steps

    def _whole_loop_applies(self, desc):
        return (self._whole_loop_ok and not self.kernel_timing and not getattr(self, "profile_step_kernel", False)
                and not self._shortcut_applies(desc)
                and bool(lib.sc_hk_run_supported(desc, self._hk, self._ovl_t0)))

    # run() with a constant dense Hessian: from this many steps on the loop runs in normal-mode coordinates (sc_hk_run_modal; the two
    # changes of basis of the monodromy blocks cost about as much as four steps of the product with Phi)
    normal_modes_from = 16

    def _modal_constants(self, potential, desc, dt):
        """transformed prefactor constants and per-mode step matrices for sc_hk_run_modal, or None where it does not apply"""
        if (desc.kind != _lib.SC_POT_HARMONIC_DENSE or self._pre.diag or not hasattr(potential, "_normal_modes")
                or not self._hk.real_lr or not lib.sc_hk_run_modal_supported(desc, self._hk, self._ovl_t0)):
            return None
        cache = self.__dict__.setdefault("_modal_cache", {})
        key = (float(dt), potential.hess0.numpy().tobytes(), potential._masses.numpy().tobytes())      # the VALUES: edited in place = new key
        hit = cache.get(key)
        if hit is not None:
            return hit
        A, B, Ainv, Binv, phi = potential._normal_modes(dt)
        dev = self.device
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        L1, L2, R1, R2 = (m.real.numpy() for m in (self._pre.L1, self._pre.L2, self._pre.R1, self._pre.R2))
        consts = [t((L1 @ A).astype(np.complex128)), t((L2 @ B).astype(np.complex128)),
                  t((Ainv @ R1).astype(np.complex128)), t((Binv @ R2).astype(np.complex128))]
        hk = sc_hk_consts(dim=self.dim, dprime=self._pre.dprime, diag=0, real_lr=1,
                          L1=ptr(consts[0]), L2=ptr(consts[1]), R1=ptr(consts[2]), R2=ptr(consts[3]))
        hit = {"hk": hk, "consts": consts, "phi": t(phi), "A": t(A), "B": t(B), "Ainv": t(Ainv), "Binv": t(Binv)}
        cache.clear()
        cache[key] = hit
        return hit

    def _to_normal_modes(self, modal, forward):
        """monodromy blocks <-> normal-mode coordinates, in place: Mqq~ = A^-1 Mqq A, Mqp~ = A^-1 Mqp B, Mpq~ = B^-1 Mpq A, Mpp~ = B^-1 Mpp B"""
        if "stacks" not in modal:
            A, B, Ai, Bi = modal["A"], modal["B"], modal["Ainv"], modal["Binv"]
            modal["stacks"] = {True: (torch.stack((Ai, Ai, Bi, Bi)).contiguous(), torch.stack((A, B, A, B)).contiguous()),
                               False: (torch.stack((A, A, B, B)).contiguous(), torch.stack((Ai, Bi, Ai, Bi)).contiguous())}
        left, right = modal["stacks"][forward]
        check(lib.sc_mono_similarity(self._state, ptr(left), ptr(right), self._stream()))

    def _run_whole_loop(self, desc, dt, nt, slots, potential=None):
        if desc.kind not in (_lib.SC_POT_MORSE, _lib.SC_POT_HARMONIC_SEP, _lib.SC_POT_EPS_MORSE):
            self._blocks_structurally_diagonal = False
        self._sync_dense_mono(leave_diagonal=True)
        self._set_mono_layout(_lib.SC_MONO_ROWMAJOR)
        nslots = lib.sc_hk_run_slots(self.ntraj, self.dim)
        # per-step partial sums of every wavefront: 40 nslots bytes per step; long runs go in launches of <= 512 steps
        # (<= 84 MB of partials at 4096 slots, reused: the launches are ordered on the stream), the state stays on the device
        # in between and is read / written once per launch (<= 1 KB per trajectory at D <= 12: negligible against 512 steps)
        chunk = min(nt, 512)
        partials = torch.empty((chunk, nslots, 5), dtype=F64, device=self.device)
        nac = self._nac
        modal = self._modal_constants(potential, desc, dt) if nt >= self.normal_modes_from else None
        if modal is not None:
            self._to_normal_modes(modal, forward=True)
        for k0 in range(0, nt, chunk):
            k = min(chunk, nt - k0)
            tail = (ptr(partials), slots.data_ptr() + 40 * k0, ptr(self._elog), self._stream())
            head = (self._ovl_t0, nac, ptr(self._vi), ptr(self.probi), ptr(self._nacq) if nac is not None else None, self._mc_norm(), dt, k)
            if modal is not None:
                check(lib.sc_hk_run_modal(desc, self._state, modal["hk"], *head, ptr(modal["phi"]), *tail))
            else:
                check(lib.sc_hk_run(desc, self._state, self._hk, *head, *tail))
        if modal is not None:
            self._to_normal_modes(modal, forward=False)
        self._run_scratch = partials          # alive until the stream has consumed it
        self._nsteps += nt
        for _ in range(nt):
            self.t += dt                      # accumulated as the reference's loop does (propagators.py:655)

    _multi_ok = True                # WM needs its own kernel after every single step
    pair_steps = True               # run(): two time steps per visit where sc_hk_step_multi applies (False: one launch per step)

    def _multi_applies(self, desc):
        """two time steps per visit (sc_hk_step_multi): separable potential, diagonal widths, 16 < D <= 64 (the tiled fast path),
        and blocks that are still diagonal -- an intermediate determinant with a weak pivot could not be repaired"""
        if not (self._multi_ok and self.pair_steps and self._blocks_structurally_diagonal and self._nac_generic is None
                and not self._shortcut_applies(desc) and self._fast_path_layout(desc) == _lib.SC_MONO_TILED16):
            return False
        probe = sc_state.from_buffer_copy(self._state)
        probe.mono_layout = _lib.SC_MONO_TILED16
        return bool(lib.sc_hk_step_multi_supported(desc, probe, self._hk))

    def _launch_step_pair(self, desc, dt):
        n, d, dev = self.ntraj, self.dim, self.device
        if self._multi is None:
            grid = self._gstep
            bufs = {"work": torch.empty((2, n, 4, d), dtype=F64, device=dev), "qp": torch.empty((n, 2 * d), dtype=F64, device=dev),
                    "act": torch.empty(n, dtype=F64, device=dev), "c2": torch.empty(n, dtype=C128, device=dev),
                    "sgn": torch.empty(n, dtype=F64, device=dev), "bad": torch.zeros(1, dtype=torch.int32, device=dev),
                    "epart": torch.zeros((2, grid), dtype=F64, device=dev)}
            bufs["ms"] = sc_multi_scratch(work=ptr(bufs["work"]), qp_mid=ptr(bufs["qp"]), act_mid=ptr(bufs["act"]), c2_mid=ptr(bufs["c2"]),
                                          sgn_mid=ptr(bufs["sgn"]), unrepaired=ptr(bufs["bad"]))
            mid = sc_state.from_buffer_copy(self._state)
            mid.qp, mid.act, mid.c2, mid.sgn = ptr(bufs["qp"]), ptr(bufs["act"]), ptr(bufs["c2"]), ptr(bufs["sgn"])
            bufs["state_mid"] = mid
            self._multi = bufs
        m = self._multi
        s = self._stream()
        self._sync_dense_mono(leave_diagonal=True)
        self._set_mono_layout(_lib.SC_MONO_TILED16)
        m["state_mid"].mono_layout = _lib.SC_MONO_TILED16
        timed = getattr(self, "profile_step_kernel", False)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        with self._timed("hk_step_pair"):
            check(lib.sc_hk_step_multi(desc, self._state, self._hk, m["ms"], dt, ptr(m["epart"]), s))
        if timed:
            e1.record()
            self.__dict__.setdefault("_step_events", []).append((e0, e1, 2))
        for sub in range(2):
            check(lib.sc_energy_guard(C_void(m["epart"].data_ptr() + 8 * sub * self._gstep), self._gstep, float(n), ptr(self._elog), s))
        self._nsteps += 2

    def _run_graph(self, potential, dt, nt, desc, slots):
        """first iteration eagerly (lazy set-up, layout conversion), then one captured iteration replayed nt - 1 times"""
        base = slots.data_ptr()
        cursor = torch.zeros(1, dtype=torch.int64, device=self.device)

        def iteration():
            self._launch_correlate(base, per_trajectory=False, cursor=cursor)
            self._launch_step(potential, dt, desc=desc, remembered=True)
        iteration()
        graph = torch.cuda.CUDAGraph()
        steps_before = self._nsteps
        with torch.cuda.graph(graph):
            iteration()                      # recorded, not executed
        self._nsteps = steps_before + 1      # the host-side bookkeeping of a captured iteration ran once ...
        graph.replay()
        for _ in range(nt - 2):
            graph.replay()
            self._nsteps += 1                # ... and is repeated here for every further replay
        for _ in range(nt):
            self.t += dt                      # accumulated as the reference's loop does (propagators.py:655)
        if hasattr(self, "_wm_step"):
            self._wm_step = self._nsteps      # the WM terms of the current state were produced by the last replay
        self._graph = graph                  # keeps the captured launch parameters alive until the work has run

    def _check_slots(self, slots, nt):
        """the kernels write 5 doubles at slots + 40 k for k < nt: refuse anything that is not exactly that buffer"""
        if not (isinstance(slots, torch.Tensor) and slots.dtype == F64 and slots.dim() == 2 and slots.shape[1] == 5
                and slots.shape[0] >= nt and slots.is_contiguous() and slots.device == self.device):
            raise ValueError(f"slots has to be a contiguous float64 tensor of shape (>= {nt}, 5) on {self.device}, got "
                             f"{getattr(slots, 'dtype', type(slots))} {tuple(getattr(slots, 'shape', ()))} on "
                             f"{getattr(slots, 'device', '?')}")

    @staticmethod
    def finalize_slots(slots, t0, dt, energy0_es):
        """apply the dynamical phase e^{i t E0/hbar} (reference propagators.py:841, 906) on the host"""
        raw = slots.detach().cpu().numpy()
        times = t0 + hostmath.time_grid(raw.shape[0], dt)
        phase = np.exp(1j / hbar * times * energy0_es)
        return (raw[:, 0] + 1j * raw[:, 1]) * phase, (raw[:, 2] + 1j * raw[:, 3]) * phase

    # ------------------------------------------------------------------ data access (reference :914-948)
    @property
    def y(self):
        """state in the reference's layout: rows (q, p, Mqq, Mqp, Mpq, Mpp, S), trajectories fastest"""
        d, n = self.dim, self.ntraj
        out = torch.empty((2 * d + 4 * d * d + 1, n), dtype=F64, device=self.device)
        self._sync_dense_mono()
        self._set_mono_layout(_lib.SC_MONO_ROWMAJOR)
        check(lib.sc_state_to_reference(self._state, ptr(out), self._stream()))
        return out

    @y.setter
    def y(self, value):
        value = torch.as_tensor(value, dtype=F64).to(self.device).contiguous()
        d, n = self.dim, self.ntraj
        assert value.shape == (2 * d + 4 * d * d + 1, n)
        self._state.mono_layout = _lib.SC_MONO_ROWMAJOR          # everything in mono is overwritten
        check(lib.sc_state_from_reference(ptr(value), self._state, self._stream()))
        torch.cuda.current_stream(self.device).synchronize()     # `value` may be a temporary
        self._corr_step = -1
        self._wm_export_step = -1
        # the new monodromy blocks may or may not be diagonal: keep the shortcut only if they are
        diag = torch.diagonal(self._mono, dim1=2, dim2=3)
        self._mono_stale = False
        self._mono_is_diag = bool(torch.count_nonzero(self._mono) == torch.count_nonzero(diag))
        self._blocks_structurally_diagonal = self._mono_is_diag
        if self._mono_is_diag:
            self._mdiag.copy_(diag)

    @property
    def c(self):
        """unsigned prefactor sqrt(c2) (principal branch), as the reference's attribute"""
        return torch.sqrt(self._c2)

    def initial_positions_and_momenta(self):
        return torch.split(self.zi, [self.dim, self.dim])

    def current_positions_and_momenta(self):
        qp = self._qp.t()
        return qp[:self.dim], qp[self.dim:]

    def classical_action(self):
        return self._act

    def monodromy_matrices(self):
        # (n, D, D) -> (D, D, n) views
        self._sync_dense_mono()
        self._set_mono_layout(_lib.SC_MONO_ROWMAJOR)
        return tuple(self._mono[:, k].permute(1, 2, 0) for k in range(4))

    def semiclassical_prefactor(self):
        return self._sgn * torch.sqrt(self._c2)

    def coefficients(self):
        """expansion coefficients v_i of the wavefunction in the coherent states (reference propagators.py:657-686)"""
        v = self.semiclassical_prefactor() * torch.exp(1j / hbar * self._act) * self._vi
        return v / (self._mc_norm() * self.probi)

    def _norm_group(self, across_ranks, group):
        """process group of a cross-rank norm(), or the LOCAL sentinel.  Cross-rank is OPT-IN: a process group that merely
        exists (ranks propagating independent ensembles, rank-0-only logging) must not turn norm() into a collective."""
        from . import distributed as Dm
        if not across_ranks and group is None:
            return Dm.LOCAL
        if Dm._active(group):
            # every rank has to hold one shard of ONE ensemble weighted with the global N (initial_conditions(ntraj_total=...))
            count = torch.tensor([float(self.ntraj)], dtype=F64, device=self.device)
            Dm.all_reduce_sum(count, group)
            if int(round(float(count.item()))) != int(self._ntraj_norm):
                raise ValueError(f"norm(across_ranks=True): the ranks hold {int(count.item())} trajectories together but the "
                                 f"Monte-Carlo weight is 1/{self._ntraj_norm}: the ranks do not share one ensemble "
                                 "(pass ntraj_total = the global count to initial_conditions, or take the rank-local norm())")
        return group

    def norm(self, across_ranks=False, group=None):
        """|psi| = sqrt(sum_ij v_i^* <q_i,p_i,Gamma_t|q_j,p_j,Gamma_t> v_j), O(n^2) (reference propagators.py:734-782).

        The pair sum runs in ``sc_pair_sum_rect``; the host only packs the operands: positions are centred (the overlaps
        depend on differences) so that the per-trajectory and the cross terms of the exponent stay small.

        By default the norm of THIS rank's trajectories with this rank's weights (no communication).
        ``across_ranks=True`` (or an explicit ``group``) -- every rank of the group holds one shard of one ensemble with
        the global N as ``ntraj_total`` and every rank makes the call: the ket operands of all ranks are all-gathered,
        every rank sums ITS bras against ALL kets and one all-reduce adds the partial sums -- every rank returns the norm
        of the whole wavefunction.
        """
        from . import distributed as Dm
        group = self._norm_group(across_ranks, group)
        dev, d, n = self.device, self.dim, self.ntraj
        oc = hostmath.OverlapConstants(self._Gt, self._Gt)
        A, B, Cm = (m.to(dev) for m in (oc.A, oc.B, oc.C))
        # a centre all ranks agree on: the mean position of the whole ensemble
        qsum = torch.cat((self._qp[:, :d].sum(0), torch.tensor([float(n)], dtype=F64, device=dev)))
        Dm.all_reduce_sum(qsum, group)
        q = self._qp[:, :d] - (qsum[:d] / qsum[d]).unsqueeze(0)
        p = self._qp[:, d:]
        Aq, Bp, Cp = q @ A.T, p @ B.T / hbar ** 2, p @ Cm.T / hbar
        # exponent_ij = rs_i + rs_j + X1_i.Y1_j + i (ib_i + ik_j + X2_i.Y2_j)   (propagators.py:232-237 expanded)
        X1 = torch.cat((q, p), 1).contiguous()
        Y1 = torch.cat((Aq, Bp), 1).contiguous()
        X2 = torch.cat((q, Cp, q), 1).contiguous()
        Y2 = torch.cat((p / hbar, -q, -Cp), 1).contiguous()
        rs = (-0.5 * (q * Aq).sum(1) - 0.5 * (p * Bp).sum(1)).contiguous()
        cc = (q * Cp).sum(1)
        ib = cc.contiguous()
        ik = (cc - (p * q).sum(1) / hbar).contiguous()
        v = self.coefficients()
        wb, wk = (oc.fac * v.conj()).contiguous(), v.contiguous()
        # kets of the whole ensemble (this rank's own when there is one rank)
        ket = torch.cat((Y1, Y2, rs.unsqueeze(1), ik.unsqueeze(1), torch.view_as_real(wk)), 1)
        ket = Dm.all_gather_rows(ket, group)
        nj = ket.shape[0]
        Y1j, Y2j = ket[:, :2 * d].contiguous(), ket[:, 2 * d:5 * d].contiguous()
        rsj, ikj = ket[:, 5 * d].contiguous(), ket[:, 5 * d + 1].contiguous()
        wkj = ket[:, 5 * d + 2:5 * d + 4].contiguous()
        tiles = lib.sc_pair_sum_rect_tiles(n, nj)
        partials = torch.empty((tiles, 4), dtype=F64, device=dev)
        slot = torch.zeros(8, dtype=F64, device=dev)
        s = self._stream()
        check(lib.sc_pair_sum_rect(ptr(X1), ptr(Y1j), 2 * d, ptr(X2), ptr(Y2j), 3 * d, ptr(rs), ptr(rsj), ptr(ib), ptr(ikj),
                                   ptr(wb), ptr(wkj), n, nj, ptr(partials), s))
        check(lib.sc_reduce_slot(ptr(partials), int(tiles), None, 0, 1.0, ptr(slot), s))
        Dm.all_reduce_sum(slot, group)
        return float(torch.sqrt(slot[0]).item())

    def wavefunction(self, x):
        """frozen-Gaussian wavefunction psi(x,t) on a spatial grid x (dim,nx) -> complex ndarray (nx,)
        (reference propagators.py:688-732 with CoherentStatesWavefunction :243-292)"""
        x = torch.as_tensor(x, dtype=F64)
        d, nx = x.shape
        assert d == self.dim, "spatial grid has wrong dimensions"
        dev, n = self.device, self.ntraj
        L = hostmath.psd_sqrt_real(self._Gt)                  # (x-q)^T Gt (x-q) = |L (x-q)|^2
        fac = hostmath.wavepacket_norm_factor(self._Gt)
        Ld = L.to(dev)
        q, p = self._qp[:, :d], self._qp[:, d:]
        LqT = (Ld @ q.T).contiguous()
        PT = p.T.contiguous()
        pq = ((p * q).sum(1) / hbar).contiguous()
        if hbar != 1.0:
            PT = PT / hbar
        X = x.T.contiguous().to(dev)
        Lx = (X @ Ld.T).contiguous()
        v = self.coefficients().contiguous()
        phi = torch.zeros(nx, dtype=C128, device=dev)
        check(lib.sc_grid_sum(ptr(LqT), ptr(PT), ptr(pq), ptr(v), n, d, ptr(Lx), ptr(X), nx, fac, ptr(phi),
                              self._stream()))
        return phi.cpu().numpy()

    def _get_signs_of_sqrt(self, key):
        if key != "prefactorC":
            logger.error(f"Apparently the sign of the square root of the quantity '{key}' is not being tracked.")
            raise KeyError(key)
        return self._sgn.type(C128)

    _TRACKED = {"prefactorC": ("_sgn", "_c2")}

    @property
    def sign_trackers(self):
        """the reference's bookkeeping dict (propagators.py:1006-1066): name -> {'signs': +-1 per trajectory,
        'previous': value at the last tracked step}, built from the device-resident tracker state (read-only views)"""
        return {name: {"signs": getattr(self, s).type(C128), "previous": getattr(self, z)}
                for name, (s, z) in self._TRACKED.items()}


class _KernelTimer(object):
    def __init__(self, prop, label):
        self.prop, self.label = prop, label

    def __enter__(self):
        self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.e0.record()

    def __exit__(self, *exc):
        self.e1.record()
        self.prop.__dict__.setdefault("_kernel_events", {}).setdefault(self.label, []).append((self.e0, self.e1))
        return False


class _NoTimer(object):
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_TIMER = _NoTimer()


def C_void(address):
    import ctypes
    return ctypes.c_void_p(int(address))


def _NULL_POT(dim):
    """descriptor for prefactor-only launches (the potential is not touched in mode 1)"""
    return _lib.sc_potential(kind=_lib.SC_POT_HARMONIC_SEP, dim=dim)


class WaltonManolopoulosPropagator(HermanKlukPropagator):
    """Walton-Manolopoulos (Filinov-smoothed) propagator, reference propagators.py:1077-1719.

    ``alpha``, ``beta``: widths of the phase-space cell over which the HK integrand is integrated out.
    The Filinov matrix, its inverse / determinant, the second inverse / determinant and the terms of
    eqns (85) and (100) are evaluated per trajectory by ``sc_wm_correlate`` right after the HK step kernel
    (registers for small matrices, LDS for medium ones, an L2-resident scratch block beyond that: no size limit).
    """

    _tiled_fast_path = False        # the Filinov matrix is built from the row-major blocks after every step
    _whole_loop_ok = False          # ... by its own kernel, between the steps
    _multi_ok = False

    def __init__(self, Gamma_i, Gamma_t, alpha, beta, device='cuda'):
        super().__init__(Gamma_i, Gamma_t, device=device)      # the Filinov matrix needs the dense monodromy blocks
        self.alpha = torch.tensor(float(alpha))
        self.beta = torch.tensor(float(beta))

    def _prepare(self):
        super()._prepare()
        dev, n = self.device, self.ntraj
        wm = hostmath.WMConstants(self._G0h, self._Gi, self._Gt, self._iGi0h, self.U.cpu(),
                                  float(self.alpha), float(self.beta))
        self._wm_host = wm
        up = lambda x: x.contiguous().to(dev)
        self._wm_bufs = {k: up(getattr(wm, k)) for k in ("U", "Gt", "G0", "iGi0", "S", "Cqq", "Cst", "Bq")}
        self._detA = torch.ones(n, dtype=C128, device=dev)
        self._detM = torch.ones(n, dtype=C128, device=dev)
        self._sgnA = torch.ones(n, dtype=F64, device=dev)
        self._sgnM = torch.ones(n, dtype=F64, device=dev)
        self._wm_flags = torch.zeros(n + 1, dtype=torch.int32, device=dev)      # fixed-order pivots handed to the pivoted kernel
        self._gwm = lib.sc_wm_grid(n, self.dim)
        self._wpart = torch.zeros((self._gwm, 4), dtype=F64, device=dev)
        self._wm_step, self._wm_has_nac = -1, False
        self._wm_export_step = -1
        self._wm_nac_bufs, self._wm_nac_traj = None, None
        need = lib.sc_wm_scratch_bytes(n, self.dim, wm.dprime)      # 0 while the matrices of a trajectory fit on chip
        assert need >= 0
        self._wm_scratch = torch.empty(need // 8, dtype=F64, device=dev) if need > 0 else None
        self._build_wm_struct()

    def _build_wm_struct(self):
        b, wm, nb = self._wm_bufs, self._wm_host, self._wm_nac_bufs
        self._wm = sc_wm_consts(
            dim=self.dim, dprime=wm.dprime, U=ptr(b["U"]), Gt=ptr(b["Gt"]), G0=ptr(b["G0"]), iGi0=ptr(b["iGi0"]),
            S=ptr(b["S"]), Cqq=ptr(b["Cqq"]), Cst=ptr(b["Cst"]), Bq=ptr(b["Bq"]), q0=ptr(self.q0), p0=ptr(self.p0),
            n1=ptr(nb[0]) if nb else None, s_n1=ptr(nb[1]) if nb else None, w_n1=ptr(nb[2]) if nb else None,
            inv_scale_a=wm.inv_scale_a, inv_two_pi=wm.inv_two_pi, pre=wm.pre,
            p0n1=self._wm_p0n1 if nb else 0.0, n2=self._wm_n2 if nb else 0.0,
            detA=ptr(self._detA), detM=ptr(self._detM), sgnA=ptr(self._sgnA), sgnM=ptr(self._sgnM),
            pre_coef=wm.pre_coef, scratch=ptr(self._wm_scratch),
            scratch_bytes=0 if self._wm_scratch is None else self._wm_scratch.numel() * 8, flags=ptr(self._wm_flags),
            nac_traj=ptr(getattr(self, "_wm_nac_traj", None)))

    def _remember_nac(self, potential):
        current, fp = self._nac_is_current(potential)
        if current:
            return
        super()._remember_nac(potential)
        masses, tau1, tau2, varies = fp
        wm = self._wm_host
        dev = self.device
        if bool(varies.item()):
            # position-dependent couplings: per trajectory n1(q_i), S n1(q_i), G0 n1(Q), p0.n1(Q), n2(q_i), n2(Q)
            # (sc_wm_consts.nac_traj); the q_i part is filled once, the Q part before every launch (_refresh_nac_traj)
            d, n = self.dim, self.ntraj
            self._wm_S, self._wm_G0 = wm.S.to(dev), wm.G0.to(dev)
            blk = torch.zeros((n, 3 * d + 3), dtype=F64, device=dev)
            n1q, n2q = self._coupling_vectors(potential, self.zi[:d])
            blk[:, :d] = n1q.t()
            blk[:, d:2 * d] = (self._wm_S @ n1q).t()
            blk[:, 3 * d + 1] = n2q
            self._wm_nac_traj = blk
            self._wm_nac_bufs = None
            self._build_wm_struct()
            return
        self._wm_nac_traj = None
        n1 = -hbar ** 2 * tau1 / masses
        self._wm_nac_bufs = [n1.to(dev), (wm.S @ n1).to(dev), (wm.G0 @ n1).to(dev)]
        self._wm_p0n1 = float(torch.dot(self._p0h, n1))
        self._wm_n2 = float(-hbar ** 2 * 0.5 * torch.sum(tau2 / masses))
        self._build_wm_struct()

    def _refresh_nac_traj(self):
        """the current-point half of sc_wm_consts.nac_traj: G0 n1(Q), p0.n1(Q), n2(Q) (reference propagators.py:1689-1697)"""
        d = self.dim
        n1Q, n2Q = self._coupling_vectors(self._nac_pot, self._qp.t()[:d])
        blk = self._wm_nac_traj
        blk[:, 2 * d:3 * d] = (self._wm_G0 @ n1Q).t()
        blk[:, 3 * d] = self.p0 @ n1Q
        blk[:, 3 * d + 2] = n2Q

    def _wm_launch(self, track):
        traj = getattr(self, "_wm_nac_traj", None)
        has_nac = self._wm_nac_bufs is not None or traj is not None
        if traj is not None:
            self._refresh_nac_traj()
        with self._timed("wm"):
            check(lib.sc_wm_correlate(self._state, self._wm, ptr(self._zi_t), ptr(self.probi), self._mc_norm(),
                                      track, int(has_nac), ptr(self._cq), ptr(self._kq), ptr(self._wpart),
                                      self._stream()))
        self._wm_step, self._wm_has_nac = self._nsteps, has_nac

    def _after_prefactor(self, track):
        self._wm_launch(track)

    def _export(self):
        """per-trajectory v_n, C_QQ and d-vector of the current step (no tracking), reference :1391-1432, 1513"""
        if getattr(self, "_wm_export_step", -1) == self._nsteps:
            return self._wm_export
        dev, n, d = self.device, self.ntraj, self.dim
        coef = torch.zeros(n, dtype=C128, device=dev)
        cqq = torch.zeros((n, d, d), dtype=C128, device=dev)
        dvec = torch.zeros((n, d), dtype=C128, device=dev)
        self._wm.coef_out, self._wm.cqq_out, self._wm.dvec_out = ptr(coef), ptr(cqq), ptr(dvec)
        try:
            self._wm_launch(0)
        finally:
            self._wm.coef_out, self._wm.cqq_out, self._wm.dvec_out = None, None, None
        self._wm_export, self._wm_export_step = (coef, cqq, dvec), self._nsteps
        return self._wm_export

    def coefficients(self):
        """coefficients v_n of the Gaussians in the WM wavefunction, eqn (75) (reference propagators.py:1391-1432)"""
        return self._export()[0]

    def wavefunction(self, x):
        """WM wavefunction psi(x,t) on a spatial grid x (dim,nx) -> complex ndarray (nx,) (reference :1434-1482)"""
        x = torch.as_tensor(x, dtype=F64)
        d, nx = x.shape
        assert d == self.dim, "spatial grid has wrong dimensions"
        coef, cqq, dvec = self._export()
        X = x.T.contiguous().to(self.device)
        phi = torch.zeros(nx, dtype=C128, device=self.device)
        check(lib.sc_wm_grid_sum(ptr(self._qp), ptr(coef), ptr(cqq), ptr(dvec), self.ntraj, d, ptr(X), nx, ptr(phi),
                                 self._stream()))
        return phi.cpu().numpy()

    def norm(self, across_ranks=False, group=None):
        """norm |psi| of the WM wavefunction, O(n^2) with a d' x d' inverse per pair (reference :1484-1575); rank-local by
        default, across ranks (opt-in) as HermanKlukPropagator.norm: every rank sums its bras against the all-gathered
        kets, one all-reduce"""
        from . import distributed as Dm
        group = self._norm_group(across_ranks, group)
        dev, n, d = self.device, self.ntraj, self.dim
        coef, cqq, dvec = self._export()
        U = self._wm_bufs["U"]                                   # (D, d') real
        Uc = U.type(C128)
        dp = U.shape[1]
        cqqp = torch.einsum('ak,nab,bl->nkl', Uc, cqq, Uc).contiguous()      # U^T CQQ U per trajectory
        dvecp = (dvec @ Uc).contiguous()
        flat = lambda t: torch.view_as_real(t.contiguous()).reshape(t.shape[0], -1)
        ket = torch.cat((self._qp, flat(coef.unsqueeze(1)), flat(cqq), flat(dvec), flat(cqqp), flat(dvecp)), 1)
        ket = Dm.all_gather_rows(ket, group)
        nj = ket.shape[0]
        widths = [2 * d, 2, 2 * d * d, 2 * d, 2 * dp * dp, 2 * dp]
        qpj, coefj, cqqj, dvecj, cqqpj, dvecpj = (c.contiguous() for c in torch.split(ket, widths, dim=1))
        tiles = lib.sc_wm_pair_sum_rect_tiles(n, nj)
        partials = torch.empty((tiles, 4), dtype=F64, device=dev)
        slot = torch.zeros(8, dtype=F64, device=dev)
        s = self._stream()
        check(lib.sc_wm_pair_sum_rect(ptr(self._qp), ptr(coef), ptr(cqqp), ptr(dvecp), n, ptr(qpj), ptr(coefj), ptr(cqqj),
                                      ptr(dvecj), ptr(cqqpj), ptr(dvecpj), nj, ptr(U), d, dp, ptr(partials), s))
        check(lib.sc_reduce_slot(ptr(partials), int(tiles), None, 0, 1.0, ptr(slot), s))
        Dm.all_reduce_sum(slot, group)
        return float(torch.sqrt(slot[0]).item())

    def _launch_correlate(self, slot_ptr, per_trajectory=True, cursor=None, slot_row=None):
        # the per-trajectory terms were produced together with the prefactor; recompute (with the stored branch
        # signs, no tracking) only if the coupling vector was not known at that time
        knows_nac = self._wm_nac_bufs is not None or self._wm_nac_traj is not None
        if self._wm_step != self._nsteps or (knows_nac and not self._wm_has_nac):
            self._wm_launch(0)
        self._reduce_into(self._wpart, self._gwm, slot_ptr, cursor)

    def _correlate_current(self, need_nac):
        if self._corr_step == self._nsteps and (self._corr_has_nac or not need_nac):
            return
        self._launch_correlate(self._slot.data_ptr())
        self._corr_step, self._corr_has_nac = self._nsteps, self._wm_has_nac
        self._slot_host = self._slot.cpu().numpy().copy()
        self._check_energy_guard()

    _TRACKED = {"prefactorC": ("_sgn", "_c2"), "detA": ("_sgnA", "_detA"), "detM": ("_sgnM", "_detM")}

    def _get_signs_of_sqrt(self, key):
        if key == "detA":
            return self._sgnA.type(C128)
        if key == "detM":
            return self._sgnM.type(C128)
        return super()._get_signs_of_sqrt(key)
