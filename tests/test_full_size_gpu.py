"""BASELINE.json configs[1] at its FULL size (60-mode anharmonic AS, 10^5 trajectories): the oracle cannot run this in
seconds, so the checks are the size-independent properties the method offers -- determinism, invariance of the
correlation sums under sharding (the multi-GPU decomposition), agreement of the dense-state kernel with the
diagonal-state shortcut, C(0) = 1 within the Monte-Carlo error, silent energy guard."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_default_dtype(torch.float64)

N, NT = 100000, 4


def _setup():
    import bench
    from semiclassical_amd import potentials as P
    omega, chi, nac, q0, dt = bench.as60_model()
    return omega, P.MorsePotential(omega, chi.clone(), nac), q0, dt, torch.diag(omega), float(0.5 * omega.sum())


def _run(G, q0, pot, dt, zi, probi, ntraj_total, **kw):
    from semiclassical_amd import propagators as PR
    prop = PR.HermanKlukPropagator(G, G, device="cuda", **kw)
    prop.set_initial_conditions(q0, 0.0 * q0, G, zi, probi, ntraj_total=ntraj_total)
    slots = torch.zeros((NT, 5), device="cuda")
    prop.run(pot, dt, NT, 0.0, slots=slots)
    prop.synchronize()
    return prop, slots.cpu().numpy()


def test_full_size_properties():
    from semiclassical_amd import propagators as PR
    omega, pot, q0, dt, G, E0 = _setup()
    seed = PR.HermanKlukPropagator(G, G, device="cuda")
    seed.initial_conditions(q0, 0.0 * q0, G, ntraj=N, generator=torch.Generator().manual_seed(11))
    zi, probi = seed.zi.cpu(), seed.probi.cpu()
    del seed
    torch.cuda.empty_cache()
    full, s_full = _run(G, q0, pot, dt, zi, probi, N)
    # C(0) = <phi|phi> = 1 within the Monte-Carlo error of 10^5 samples
    assert abs(complex(s_full[0, 0], s_full[0, 1]) - 1.0) < 2e-2
    c2_full, qp_full, act_full = full._c2.clone(), full._qp.clone(), full._act.clone()
    del full
    torch.cuda.empty_cache()
    # determinism: the same inputs give the same bits
    again, s_again = _run(G, q0, pot, dt, zi, probi, N)
    assert np.array_equal(s_again, s_full) and torch.equal(again._c2, c2_full)
    del again
    torch.cuda.empty_cache()
    # sharding invariance: two ranks' worth of trajectories, each normalised with the global count, add up to the
    # single-batch sums (the only coupling between trajectories is this sum)
    h = N // 2
    _, s_a = _run(G, q0, pot, dt, zi[:, :h], probi[:h], N)
    _, s_b = _run(G, q0, pot, dt, zi[:, h:], probi[h:], N)
    assert np.max(np.abs((s_a + s_b)[:, :4] - s_full[:, :4])) < 1e-12 * max(1.0, np.max(np.abs(s_full[:, :4])))
    torch.cuda.empty_cache()
    # the dense-state kernel and the diagonal-state shortcut describe the same trajectories
    short, s_short = _run(G, q0, pot, dt, zi, probi, N, exploit_separability=True)
    assert short._mono_stale
    assert torch.equal(short._qp, qp_full) and torch.equal(short._act, act_full)
    assert float((short._c2 - c2_full).abs().max() / c2_full.abs().max()) < 1e-12
    assert np.max(np.abs(s_short[:, :4] - s_full[:, :4])) < 1e-11
