"""sGDML potential on the HIP engine (reference semiclassical/potentials.py:641-744, gdml_predictor.py).

``MolecularGDMLPotential(model_pot, nac_fchk)`` keeps the reference's constructor: ``model_pot`` is the mapping
of an sGDML ``.npz`` model (keys ``sig, c, std, z, R_desc, R_d_desc_alpha, perms, tril_perms_lin``), ``nac_fchk``
provides ``nonadiabatic_coupling()``, ``atomic_numbers()`` and ``masses()``.  Energy, gradient and the analytic
Hessian are evaluated by ``sc_gdml_eval`` / ``sc_gdml_stage`` -- there is no torch implementation of the force
field in this package.
"""
import numpy as np
import torch

from ._lib import lib, check, ptr, sc_gdml_model
from .potentials import _MolecularPotentialBase

__all__ = ['MolecularGDMLPotential']


class MolecularGDMLPotential(_MolecularPotentialBase):
    def __init__(self, model_pot, nac_fchk, device='cuda'):
        model = dict(model_pot)
        z = np.asarray(model['z'])
        assert np.array_equal(z, np.asarray(nac_fchk.atomic_numbers())), \
            "GDML models for potential energy and NAC vector should be for the same molecule."
        self.nac0 = torch.from_numpy(np.asarray(nac_fchk.nonadiabatic_coupling(), dtype=np.float64))
        self._masses = torch.from_numpy(np.asarray(nac_fchk.masses(), dtype=np.float64))
        self._dim = len(self._masses)
        # training data expanded over the permutations, gdml_predictor.py:66-85
        desc = np.asarray(model['R_desc']).shape[0]
        n_perms, n_atoms = np.asarray(model['perms']).shape
        perm = torch.tensor(np.asarray(model['tril_perms_lin'])).view(-1, n_perms).t()
        expand = lambda xs: xs.repeat(1, n_perms)[:, perm].reshape(-1, desc).contiguous()
        self._xs_train = expand(torch.tensor(np.asarray(model['R_desc'], dtype=np.float64)).t())
        self._jx_alphas = expand(torch.tensor(np.asarray(model['R_d_desc_alpha'], dtype=np.float64)))
        self._sig, self._c, self._std = int(model['sig']), float(model['c']), float(model.get('std', 1))
        self._n_atoms = int(n_atoms)
        assert lib.sc_gdml_row_len(self._n_atoms) > 0, "the sGDML kernels hold molecules of up to 48 atoms"
        k, l = torch.tril_indices(n_atoms, n_atoms, offset=-1)
        self._pair_k, self._pair_l = k.to(torch.int32).contiguous(), l.to(torch.int32).contiguous()
        self._default_device = device
        self._model_cache = {}

    def _invalidate_descriptor(self):
        self._model_cache = {}

    def _gdml_model(self, device):
        """the ``sc_gdml_model`` parameter block on ``device`` (device buffers are cached per device)"""
        key = str(device)
        if key not in self._model_cache:
            up = lambda t: t.to(device)
            bufs = [up(self._xs_train), up(self._jx_alphas), up(self._pair_k), up(self._pair_l), up(1.0 / self._masses)]
            m = sc_gdml_model(n_atoms=self._n_atoms, n_desc=self._xs_train.shape[1], n_train=self._xs_train.shape[0],
                              xs_train=ptr(bufs[0]), jx_alphas=ptr(bufs[1]), pair_k=ptr(bufs[2]), pair_l=ptr(bufs[3]),
                              q=float(np.sqrt(5) / self._sig), c=self._c, std=self._std, origin=float(self._origin),
                              inv_mass=ptr(bufs[4]))
            self._model_cache[key] = (m, bufs)
        return self._model_cache[key][0]

    def harmonic_approximation(self, r):
        """V (n,), grad (D, n), hess (D, D, n) for positions r (D, n) -- evaluated on the GPU"""
        dev = r.device if r.is_cuda else torch.device(self._default_device)
        rt = r.to(dev, torch.float64).t().contiguous()                       # (n, 3N)
        n, D = rt.shape
        e = torch.empty(n, dtype=torch.float64, device=dev)
        g = torch.empty((n, D), dtype=torch.float64, device=dev)
        h = torch.empty((n, D, D), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            check(lib.sc_gdml_eval(self._gdml_model(dev), ptr(rt), n, ptr(e), ptr(g), ptr(h),
                                   torch.cuda.current_stream(dev).cuda_stream))
        out = (e, g.t(), h.permute(1, 2, 0))
        return tuple(x.to(r.device) for x in out)

    def derivative_coupling_1st(self, r):
        return self.nac0.to(r.device).unsqueeze(1).expand_as(r)

    def derivative_coupling_2nd(self, r):
        return torch.zeros_like(r)
