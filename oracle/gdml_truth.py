"""Extended-precision evaluation of the sGDML formulas (test infrastructure, like everything under oracle/).

The sGDML energy / gradient / Hessian are sums over the training points in which terms of magnitude 2e8 cancel down
to 6e1 (coumarin model): in fp64 ANY summation order carries a rounding error of ~1e-8 relative -- the reference's own
output moves by 2e-8 when its training points are permuted (tests/test_oracle.py::test_gdml_sum_conditioning).  To
judge an fp64 implementation against something better than another fp64 implementation, this module evaluates the
same formulas (reference semiclassical/gdml_predictor.py:96-250, restated in oracle/sc_oracle.py::GDMLOracle) in
x87 extended precision (64-bit mantissa, eps = 1.1e-19), which resolves the cancellation three digits further.
Only NumPy; a handful of geometries at a time.
"""
import numpy as np

LD = np.longdouble


def forward_longdouble(model, r):
    """E, dE/dr (B, 3N), d2E/drdr (B, 3N, 3N) in extended precision for geometries r (B, 3N); n_perms = 1 models"""
    sig, c, std = int(model['sig']), LD(float(model['c'])), LD(float(model.get('std', 1)))
    assert np.asarray(model['perms']).shape[0] == 1, "permutation-expanded models are not needed by the tests"
    q = np.sqrt(LD(5)) / LD(sig)
    xs_train = np.asarray(model['R_desc'], dtype=np.float64).T.astype(LD)            # (M, Dd)
    A = np.asarray(model['R_d_desc_alpha'], dtype=np.float64).astype(LD)             # (M, Dd)
    N = int(np.asarray(model['z']).shape[0])
    k, l = np.tril_indices(N, -1)
    Dd = len(k)
    r = np.asarray(r, dtype=np.float64).astype(LD)
    B = r.shape[0]
    pos = r.reshape(B, N, 3)
    diff = pos[:, k, :] - pos[:, l, :]
    xs = LD(1) / np.sqrt(np.sum(diff * diff, axis=-1))
    xd = xs[:, None, :] - xs_train[None]
    dist = np.sqrt(np.sum(xd * xd, axis=-1))
    XA = np.einsum('bmd,md->bm', xd, A)
    ef = q ** 4 / LD(3) * np.exp(-q * dist)
    f = ef * (LD(1) + q * dist) / q ** 2
    energy = np.einsum('bm,bm->b', f, XA) * std + c
    jac = np.zeros((B, Dd, N, 3), dtype=LD)
    idx = np.arange(Dd)
    jd = -(xs ** 3)[:, :, None] * diff
    jac[:, idx, k, :] = jd
    jac[:, idx, l, :] -= jd
    jac = jac.reshape(B, Dd, 3 * N)
    gx = np.einsum('bm,md->bd', f, A) - np.einsum('bm,bmd->bd', ef * XA, xd)
    grad = np.einsum('bd,bdx->bx', gx, jac) * std
    XJ = np.einsum('bmd,bdx->bmx', xd, jac)
    AJ = np.einsum('md,bdx->bmx', A, jac)
    JJ = np.einsum('bdx,bdy->bxy', jac, jac)
    hess = np.einsum('bm,bmx,bmy->bxy', ef * XA * q / dist, XJ, XJ)
    hess -= np.einsum('bm,bxy->bxy', ef * XA, JJ)
    hess -= np.einsum('bm,bmx,bmy->bxy', ef, AJ, XJ)
    hess -= np.einsum('bm,bmx,bmy->bxy', ef, XJ, AJ)
    T = (3 * (gx * xs ** 5)[:, :, None, None] * diff[:, :, :, None] * diff[:, :, None, :]
         - (gx * xs ** 3)[:, :, None, None] * np.eye(3, dtype=LD))
    H4 = hess.reshape(B, N, 3, N, 3)
    for d in range(Dd):
        a, b = int(k[d]), int(l[d])
        H4[:, a, :, a, :] += T[:, d]
        H4[:, b, :, b, :] += T[:, d]
        H4[:, a, :, b, :] -= T[:, d]
        H4[:, b, :, a, :] -= T[:, d]
    return energy, grad, hess * std
