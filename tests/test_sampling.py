"""On-device initial conditions (sc_sample_initial, reference propagators.py:537-566).

CPU: the NumPy restatement of the generator reproduces the published Philox4x32-10 known-answer vectors.
GPU: the kernel's deviates equal that restatement, zi / probi follow the reference's formulas from those deviates, the
ensemble does not depend on how it is split, its moments are those of N(0, 1), and C(0) = 1 within the Monte-Carlo error.
"""
import numpy as np
import pytest
import torch

from tests import philox_ref

torch.set_default_dtype(torch.float64)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 with 10 rounds
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox_ref.philox4x32_10([np.array([c], dtype=np.uint64) for c in ctr], key)
        assert tuple(int(x[0]) for x in got) == want


def _setup(D, rank_deficient=False):
    rng = np.random.default_rng(5)
    if rank_deficient:
        Q, _ = np.linalg.qr(rng.normal(size=(D, D)))
        w = np.concatenate((rng.uniform(0.5, 2.0, D - 2), np.zeros(2)))
        G = torch.from_numpy((Q * w) @ Q.T)
        G = 0.5 * (G + G.T)
    else:
        G = torch.diag(torch.from_numpy(rng.uniform(0.5, 2.0, D)))
    q0, p0 = torch.from_numpy(rng.normal(size=D)), torch.from_numpy(rng.normal(size=D))
    return G, q0, p0


@pytest.mark.gpu
@pytest.mark.parametrize("D,deficient", [(1, False), (5, False), (12, True), (60, False), (70, False)])
def test_device_deviates_and_formulas(D, deficient):
    from semiclassical_amd import hostmath, propagators as PR
    from semiclassical_amd._lib import lib, check, ptr
    G, q0, p0 = _setup(D, deficient)
    n, seed, sub, first = 777, 2024, 3 + (5 << 32), 10 ** 10 + 5        # subsequence with bits above 2^32
    prop = PR.HermanKlukPropagator(G, G, device="cuda")
    prop.initial_conditions(q0, p0, G, ntraj=n, seed=seed, subsequence=sub, first_index=first)
    U, iGi0, iLz, detLz, dp = hostmath.sampling_matrices(G, G)
    # the deviates themselves (a second launch with the same arguments: the generator is counter based)
    xi = torch.empty((n, 2 * dp), device="cuda")
    zi2, pr2 = torch.empty_like(prop._zi_t), torch.empty_like(prop.probi)
    ilz_d, z0_d = iLz.contiguous().cuda(), torch.cat((q0, p0)).cuda()      # kept alive until the launch has run
    check(lib.sc_sample_initial(prop._state, ptr(ilz_d), ptr(z0_d), dp,
                                float(detLz / (2 * np.pi) ** D), seed, sub, first, 0, ptr(zi2), ptr(pr2), ptr(xi), None))
    torch.cuda.synchronize()
    assert torch.equal(zi2, prop._zi_t) and torch.equal(pr2, prop.probi)
    want = philox_ref.deviates(seed, sub, first, n, dp)
    assert np.abs(xi.cpu().numpy() - want).max() < 1e-12          # log / sincos differ in the last bits only
    # reference formulas from the kernel's own deviates (propagators.py:542, 555)
    xi_h = xi.cpu().T
    zi = torch.cat((q0, p0)).unsqueeze(1) + torch.einsum('ji,jn->in', iLz, xi_h)
    probi = detLz / (2 * np.pi) ** D * torch.exp(-0.5 * torch.einsum('in,in->n', xi_h, xi_h))
    assert (prop.zi.cpu() - zi).abs().max() < 1e-13 * max(1.0, float(zi.abs().max()))
    assert ((prop.probi.cpu() - probi).abs() / probi).max() < 1e-13
    # state of t = 0 (propagators.py:581-603)
    y = prop.y.cpu()
    assert torch.equal(y[:2 * D], prop.zi.cpu()) and float(y[-1].abs().max()) == 0.0
    eye = torch.eye(D).reshape(-1, 1).expand(-1, n)
    zero = torch.zeros(D * D, n)
    blocks = y[2 * D:-1].reshape(4, D * D, n)
    assert torch.equal(blocks[0], eye) and torch.equal(blocks[3], eye) and torch.equal(blocks[1], zero) and torch.equal(blocks[2], zero)


@pytest.mark.gpu
def test_device_ensemble_is_independent_of_the_split_and_of_the_launch():
    from semiclassical_amd import propagators as PR
    G, q0, p0 = _setup(7)
    whole = PR.HermanKlukPropagator(G, G, device="cuda")
    whole.initial_conditions(q0, p0, G, ntraj=1000, seed=9)
    a, b = PR.HermanKlukPropagator(G, G, device="cuda"), PR.HermanKlukPropagator(G, G, device="cuda")
    a.initial_conditions(q0, p0, G, ntraj=400, ntraj_total=1000, seed=9, first_index=0)
    b.initial_conditions(q0, p0, G, ntraj=600, ntraj_total=1000, seed=9, first_index=400)
    assert torch.equal(torch.cat((a._zi_t, b._zi_t)), whole._zi_t)
    assert torch.equal(torch.cat((a.probi, b.probi)), whole.probi)
    other = PR.HermanKlukPropagator(G, G, device="cuda")
    other.initial_conditions(q0, p0, G, ntraj=1000, seed=9, subsequence=1)
    assert float((other._zi_t - whole._zi_t).abs().min()) > 0.0       # another subsequence: another ensemble
    # (seed, subsequence) pairs that collided when the subsequence was xor-ed into the key (round 3): distinct streams now
    golden = 0x9E3779B97F4A7C15
    a, b = PR.HermanKlukPropagator(G, G, device="cuda"), PR.HermanKlukPropagator(G, G, device="cuda")
    a.initial_conditions(q0, p0, G, ntraj=64, seed=9, subsequence=1)
    b.initial_conditions(q0, p0, G, ntraj=64, seed=9 ^ golden, subsequence=0)
    assert float((a._zi_t - b._zi_t).abs().min()) > 0.0
    with pytest.raises(ValueError, match="subsequence"):
        a.initial_conditions(q0, p0, G, ntraj=4, seed=1, subsequence=2 ** 56)


@pytest.mark.gpu
def test_device_ensemble_statistics_and_norm():
    from scipy import stats
    from semiclassical_amd import hostmath, potentials as P, propagators as PR
    D, n = 6, 100000
    rng = np.random.default_rng(1)
    omega = torch.from_numpy(np.sort(rng.uniform(500, 3000, D)) / 219474.63)
    S = torch.from_numpy(rng.uniform(0.05, 0.4, D))
    G = torch.diag(omega)
    q0, p0 = torch.sqrt(2 * S / omega), torch.zeros(D)
    prop = PR.HermanKlukPropagator(G, G, device="cuda")
    prop.initial_conditions(q0, p0, G, ntraj=n, seed=123)
    U, iGi0, iLz, detLz, dp = hostmath.sampling_matrices(G, G)
    # xi = iLz^-T (zi - z0): moments and distribution of the recovered deviates
    dz = (prop.zi.cpu() - torch.cat((q0, p0)).unsqueeze(1))
    xi = torch.linalg.solve(iLz.T, dz).numpy()
    se = 5.0 / np.sqrt(n)
    assert np.abs(xi.mean(1)).max() < se
    cov = np.cov(xi)
    assert np.abs(cov - np.eye(2 * D)).max() < 2 * se
    assert abs(stats.kurtosis(xi.reshape(-1))) < 10 * np.sqrt(24.0 / xi.size)
    assert stats.kstest(xi.reshape(-1)[::7], "norm").pvalue > 1e-3
    # Monte-Carlo normalisation: C(0) = 1 within five standard errors (propagators.py:837; cli.py:467 asserts 1e-3)
    c_auto = prop.autocorrelation(0.0)
    cq = prop._cq.cpu().numpy()                      # the weighted per-trajectory terms C_qp / (N P (2 pi hbar)^D)
    c0, err = cq.sum(), np.sqrt(n) * cq.std()
    # (with Gamma_i = Gamma_0 the sampling density IS the integrand at t = 0: every term equals 1/N, err = 0)
    assert abs(c0 - 1.0) < max(5 * err, 1e-12) and err < 2e-3, (c0, err)
    assert abs(c_auto - c0) < 1e-12
    # and the sampled ensemble propagates like any other
    pot = P.MorsePotential(omega, torch.full((D,), 0.02), torch.from_numpy(rng.normal(0, 1e-3, D)))
    c, k = prop.run(pot, 1.5, 4, float(0.5 * omega.sum()))
    assert np.isfinite(c).all() and np.isfinite(k).all() and abs(c[0] - c0) < 1e-12
