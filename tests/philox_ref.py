"""NumPy restatement of the deviates sc_sample_initial draws (csrc/sc_sample.hip): Philox4x32-10 (Salmon, Moraes, Dror,
Shaw, SC'11), key = seed, counter = (global trajectory index lo, hi, pair index | subsequence[32..55] << 8,
subsequence[0..31]), two 53-bit uniforms per call, one Box-Muller pair.  Test infrastructure (checked against the published known-answer vectors)."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(c, k):
    """c: four uint64 arrays holding 32-bit words, k: two ints -> four arrays of 32-bit words"""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) for x in c)
    k0, k1 = k
    for _ in range(10):
        p0, p1 = c0 * np.uint64(M0), c2 * np.uint64(M1)
        hi0, lo0 = p0 >> np.uint64(32), p0 & np.uint64(MASK)
        hi1, lo1 = p1 >> np.uint64(32), p1 & np.uint64(MASK)
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c0, c1, c2, c3


def deviates(seed, subsequence, first, n, dprime):
    """xi (n, 2 d') exactly as the kernel orders them: pair p -> (xi[p], xi[d' + p])"""
    assert 0 <= subsequence < 2 ** 56 and dprime <= 256
    key = (seed & MASK, (seed >> 32) & MASK)
    g = (first + np.arange(n, dtype=np.uint64))[:, None] + np.zeros((1, dprime), dtype=np.uint64)
    p = np.zeros((n, 1), dtype=np.uint64) + np.arange(dprime, dtype=np.uint64)[None, :]
    c2 = p | np.uint64(((subsequence >> 32) << 8) & MASK)
    c3 = np.zeros_like(p) + np.uint64(subsequence & MASK)
    r0, r1, r2, r3 = philox4x32_10((g & np.uint64(MASK), g >> np.uint64(32), c2, c3), key)
    u = lambda hi, lo: ((((hi << np.uint64(32)) | lo) >> np.uint64(11)).astype(np.float64) + 1.0) / 9007199254740992.0
    u1, u2 = u(r0, r1), u(r2, r3)
    rad = np.sqrt(-2.0 * np.log(u1))
    return np.concatenate((rad * np.cos(2.0 * np.pi * u2), rad * np.sin(2.0 * np.pi * u2)), axis=1)
