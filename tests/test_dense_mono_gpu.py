"""sc_dense_mono_step (RK4 of the monodromy blocks under four dense stage Hessians on the FP64 matrix cores + prefactor)
against a plain torch fp64 restatement of the reference's equations of motion (propagators.py:86-119, 352-362) and
prefactor (:951-1004), through the C-ABI, for dimensions that cover every tile / k-slice instantiation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
torch.set_default_dtype(torch.float64)


def torch_reference(M, H, inv_mass, dt):
    """M (n,4,D,D) = Mqq, Mqp, Mpq, Mpp; H (n,4,D,D) stage Hessians -> M after one RK4 step"""
    W = inv_mass[None, :, None]

    def f(m, h):
        return torch.stack((W * m[:, 2], W * m[:, 3], -h @ m[:, 0], -h @ m[:, 1]), dim=1)
    k1 = f(M, H[:, 0])
    k2 = f(M + 0.5 * dt * k1, H[:, 1])
    k3 = f(M + 0.5 * dt * k2, H[:, 2])
    k4 = f(M + dt * k3, H[:, 3])
    return M + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)


@pytest.mark.parametrize("D", [3, 12, 16, 17, 30, 33, 45, 48, 51, 57, 61, 64, 65, 70, 80, 81, 90, 96])
def test_dense_mono_step_matches_torch(D):
    from semiclassical_amd import hostmath
    from semiclassical_amd._lib import lib, check, ptr, sc_state, sc_hk_consts
    dev, n, dt = torch.device("cuda"), 37, 0.3
    gen = torch.Generator().manual_seed(D)
    M = torch.randn((n, 4, D, D), generator=gen) * 0.3
    M[:, 0] += torch.eye(D)
    M[:, 3] += torch.eye(D)
    H = torch.randn((n, 4, D, D), generator=gen)
    H = 0.5 * (H + H.transpose(2, 3))                   # stage Hessians are symmetric (and differ per stage)
    inv_mass = 0.5 + torch.rand(D, generator=gen)
    want = torch_reference(M, H, inv_mass, dt)
    # diagonal width matrices: prefactor matrix of propagators.py:969-986 with diagonal Gamma^(+-1/2)
    gi, gt = 0.5 + torch.rand(D, generator=gen), 0.5 + torch.rand(D, generator=gen)
    st, si = torch.sqrt(gt), torch.sqrt(gi)
    mat = 0.5 * ((st[:, None] / si[None, :]) * want[:, 0] + (si[None, :] / st[:, None]) * want[:, 3]
                 - 1j * (st[:, None] * si[None, :]) * want[:, 1] + 1j * want[:, 2] / (st[:, None] * si[None, :]))
    c2_want = torch.linalg.det(mat)

    Md = M.to(dev).contiguous()
    Hd, wd = H.to(dev).contiguous(), inv_mass.to(dev)
    std, sid = st.to(dev), si.to(dev)
    c2 = torch.ones(n, dtype=torch.complex128, device=dev)
    sgn = torch.ones(n, device=dev)
    qp, act = torch.zeros((n, 2 * D), device=dev), torch.zeros(n, device=dev)
    state = sc_state(n=n, dim=D, qp=ptr(qp), act=ptr(act), mono=ptr(Md), c2=ptr(c2), sgn=ptr(sgn))
    hk = sc_hk_consts(dim=D, dprime=D, diag=1, st=ptr(std), si=ptr(sid))
    stream = torch.cuda.current_stream().cuda_stream
    sums = torch.empty_like(Md) if D > 64 else None
    check(lib.sc_dense_mono_step(state, hk, ptr(wd), ptr(Hd), ptr(sums) if D > 64 else None, dt, 0, stream))
    torch.cuda.synchronize()
    got = Md.cpu()
    assert float((got - want).abs().max() / want.abs().max()) < 1e-13
    assert float((c2.cpu() - c2_want).abs().max() / c2_want.abs().max()) < 1e-10
    assert np.all(sgn.cpu().numpy() == 1.0)               # previous c2 = 1: the branch tracker must not flip


@pytest.mark.parametrize("D", [40, 70, 96])
def test_register_prefactor_and_its_weak_pivot_fallback(D):
    """diagonal width matrices: the determinant of the dense path is taken in registers (block-pivoted elimination,
    dense_prefactor_reg_kernel); monodromy blocks whose large entries lie outside the diagonal 16-column blocks defeat the
    in-block pivoting -- those trajectories are flagged and redone by the fully pivoted LDS elimination"""
    from semiclassical_amd import propagators as PR
    from semiclassical_amd._lib import lib, check
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(D)
    n = 9
    G = torch.diag(torch.from_numpy(rng.uniform(0.5, 2.0, D)))
    q0 = torch.zeros(D)
    prop = PR.HermanKlukPropagator(G, G, device="cuda")
    prop.initial_conditions(q0, q0, G, ntraj=n, generator=torch.Generator().manual_seed(1))
    st = torch.sqrt(torch.diagonal(G)).numpy()
    for shift in (0, 20):
        blocks = []
        for _ in range(4):
            b = rng.uniform(-0.2, 0.2, (n, D, D))
            b += np.roll(np.eye(D), shift, axis=1)[None] * rng.uniform(1.0, 2.0, (n, D, 1))
            blocks.append(b)
        prop._set_mono_layout(0)
        prop._mono.copy_(torch.from_numpy(np.stack(blocks, axis=1)).cuda())
        check(lib.sc_dense_mono_step(prop._state, prop._hk, None, None, prop._mono_sums_ptr(), 0.0, 1, prop._stream()))
        torch.cuda.synchronize()
        mqq, mqp, mpq, mpp = blocks
        scale_a, scale_b = st[None, :, None], st[None, None, :]
        mat = 0.5 * (scale_a / scale_b * mqq + scale_b / scale_a * mpp - 1j * scale_a * scale_b * mqp + 1j * mpq / (scale_a * scale_b))
        want = np.linalg.det(mat)
        got = prop._c2.cpu().numpy()
        assert np.max(np.abs(got - want) / np.abs(want)) < 1e-10, shift
        flagged = int(prop._flags[-2].item())
        assert int(prop._flags[:-2].sum().item()) == 0                  # the fix-up pass clears the flags it served
        assert (flagged > 0) == (shift > 0), (shift, flagged)      # weak in-block pivots occur only for the shifted blocks
