// Initial conditions on the device: xi ~ N(0, 1) (2 d' deviates per trajectory), zi = z0 + iLz^T xi,
// P(zi) = det(Lz)/(2 pi)^D exp(-|xi|^2 / 2), and the engine state of t = 0 (q, p, S = 0, M = 1).
//                                                     (reference semiclassical/propagators.py:537-566, 581-603)
//
// The reference draws xi with torch's generator on its compute device; the stream of deviates is therefore a property
// of the torch build, not of the reference.  Here the deviates come from Philox4x32-10 (Salmon et al., SC'11: ten
// rounds, multipliers 0xD2511F53 / 0xCD9E8D57, Weyl constants 0x9E3779B9 / 0xBB67AE85):
//     key     = (seed lo, seed hi)
//     counter = (trajectory index lo, trajectory index hi, pair index | subsequence[32..55] << 8, subsequence[0..31])
// Every (seed, subsequence, trajectory, pair) is a DISTINCT Philox input (pair index < 256, subsequence < 2^56): no two
// (seed, subsequence) combinations can collide into one stream (round 3 folded the subsequence into the key by xor).
// One call gives 128 bits = two 53-bit uniforms = one Box-Muller pair (xi_j, xi_j+1).  A deviate depends only on
// (seed, subsequence, GLOBAL trajectory index, j): the ensemble is the same however the trajectories are split over
// launches or ranks (rank r passes its first global index as `first`), and ranks that pass different subsequences
// draw independent ensembles.
//
// Mapping: one wavefront per trajectory.  Lanes 0 .. d'-1 each draw one pair -> xi in LDS -> lane i computes
// zi[i], zi[i + 64], ... as dot products with the columns of iLz (L2 / LDS resident, O(D^2) constants);
// -|xi|^2/2 by a wave reduction.  Everything written once, coalesced along the trajectory's row.
#include "sc_common.h"

namespace {

constexpr int SC_SAMPLE_MAX_E = 512;      // 2 d' deviates of one trajectory staged in LDS (4 trajectories per workgroup)

struct PhiloxKey { uint32_t k0, k1; };

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, PhiloxKey key,
                                              uint32_t (&out)[4]) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    uint32_t k0 = key.k0, k1 = key.k1;
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// 53 random bits -> uniform in (0, 1]  (never 0: the logarithm of Box-Muller stays finite)
__device__ __forceinline__ double uniform_open0(uint32_t hi, uint32_t lo) {
    const uint64_t bits = (((uint64_t)hi << 32) | lo) >> 11;            // 53 bits
    return ((double)bits + 1.0) * (1.0 / 9007199254740992.0);
}

struct SampleArgs {
    sc_state st;
    const double *ilz;        // [2 d'][2 D] row-major: block_diag(iLq, iLp) of propagators.py:506-528
    const double *z0;         // [2 D]
    double *zi_t;             // [n][2 D] out
    double *probi;            // [n] out
    double *xi_out;           // [n][2 d'] out or NULL (tests: the deviates themselves)
    int dprime;
    double prob0;             // det(Lz) / (2 pi)^D
    uint64_t seed, subsequence;
    int64_t first;            // global index of this batch's trajectory 0
    int init_state;           // also write qp = zi, act = 0, mono = identity blocks (row-major), c2 = 1, sgn = 1
};

__global__ __launch_bounds__(256) void sample_initial_kernel(SampleArgs A) {
    __shared__ double xis[4][SC_SAMPLE_MAX_E];
    const int D = A.st.dim, D2 = 2 * D, dp = A.dprime, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const PhiloxKey key = {(uint32_t)A.seed, (uint32_t)(A.seed >> 32)};
    const uint32_t sub_lo = (uint32_t)A.subsequence, sub_hi = (uint32_t)(A.subsequence >> 32) << 8;
    for (int64_t tr = (int64_t)blockIdx.x * 4 + wave; tr < A.st.n; tr += (int64_t)gridDim.x * 4) {
        const uint64_t g = (uint64_t)(A.first + tr);
        double half = 0.0;
        // pair p = (xi_p, xi_{p + d'}): the position and the momentum deviate of non-zero mode p
        for (int p = lane; p < dp; p += 64) {
            uint32_t r[4];
            philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)p | sub_hi, sub_lo, key, r);
            const double u1 = uniform_open0(r[0], r[1]), u2 = uniform_open0(r[2], r[3]);
            const double rad = sqrt(-2.0 * log(u1));
            double sn, cs;
            sincospi(2.0 * u2, &sn, &cs);
            const double a = rad * cs, b = rad * sn;
            xis[wave][p] = a; xis[wave][dp + p] = b;
            half += a * a + b * b;
            if (A.xi_out) { A.xi_out[tr * 2 * dp + p] = a; A.xi_out[tr * 2 * dp + dp + p] = b; }
        }
        half = wave_sum(half);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int i = lane; i < D2; i += 64) {
            // iLz is block diagonal: positions (i < D) see the first d' deviates, momenta the last d'
            const int j0 = i < D ? 0 : dp;
            double z = A.z0[i];
            for (int j = 0; j < dp; ++j) z = fma(A.ilz[(int64_t)(j0 + j) * D2 + i], xis[wave][j0 + j], z);
            A.zi_t[tr * D2 + i] = z;
            if (A.init_state) A.st.qp[tr * D2 + i] = z;
        }
        if (lane == 0) {
            A.probi[tr] = A.prob0 * exp(-0.5 * half);
            if (A.init_state) {
                A.st.act[tr] = 0.0;
                ((cplx *)A.st.c2)[tr] = c_make(1.0, 0.0);
                A.st.sgn[tr] = 1.0;
            }
        }
        __builtin_amdgcn_wave_barrier();         // xis[wave] is rewritten by the next trajectory
    }
}

// the diagonals of Mqq and Mpp of every trajectory (the blocks were zeroed by a memset), row-major
__global__ __launch_bounds__(256) void mono_identity_kernel(double *mono, int64_t n, int D) {
    const int64_t total = n * 2 * D;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t tr = e / (2 * D);
        const int w = (int)(e - tr * 2 * D), p = w < D ? 0 : 3, a = w < D ? w : w - D;
        mono[tr * 4 * (int64_t)D * D + (int64_t)p * D * D + (int64_t)a * D + a] = 1.0;
    }
}

}  // namespace

extern "C" int sc_sample_initial(const sc_state *st, const double *ilz, const double *z0, int32_t dprime, double prob0,
                                 uint64_t seed, uint64_t subsequence, int64_t first, int32_t init_state, double *zi_t,
                                 double *probi, double *xi_out, void *stream) {
    if (!st || !ilz || !z0 || !zi_t || !probi) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_sample_initial: null argument");
    if (st->n < 0 || st->dim < 1 || dprime < 1 || dprime > st->dim)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_sample_initial: n = %lld, D = %d, d' = %d", (long long)st->n, st->dim, dprime);
    if (subsequence >> 56)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_sample_initial: subsequence %llu does not fit the counter (< 2^56)",
                       (unsigned long long)subsequence);
    if (2 * dprime > SC_SAMPLE_MAX_E)
        return sc_fail(SC_ERR_UNSUPPORTED, "sc_sample_initial: d' = %d (at most %d non-zero modes are sampled on the device; "
                       "sample on the host and use set_initial_conditions)", dprime, SC_SAMPLE_MAX_E / 2);
    if (init_state && (!st->qp || !st->act || !st->mono || !st->c2 || !st->sgn))
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_sample_initial: init_state needs the state buffers");
    if (st->n == 0) return SC_OK;
    hipStream_t s = (hipStream_t)stream;
    SampleArgs a;
    a.st = *st; a.ilz = ilz; a.z0 = z0; a.zi_t = zi_t; a.probi = probi; a.xi_out = xi_out; a.dprime = dprime;
    a.prob0 = prob0; a.seed = seed; a.subsequence = subsequence; a.first = first; a.init_state = init_state;
    const int64_t quads = (st->n + 3) / 4;
    const int grid = (int)(quads < 4096 ? quads : 4096);
    hipLaunchKernelGGL(sample_initial_kernel, dim3(grid), dim3(256), 0, s, a);
    int rc = sc_check_launch("sc_sample_initial");
    if (rc != SC_OK || !init_state) return rc;
    if (hipMemsetAsync(st->mono, 0, sizeof(double) * 4 * (size_t)st->dim * st->dim * (size_t)st->n, s) != hipSuccess)
        return sc_check_launch("sc_sample_initial (memset)");
    const int64_t diag_blocks = (st->n * 2 * st->dim + 255) / 256;
    hipLaunchKernelGGL(mono_identity_kernel, dim3((int)(diag_blocks < 8192 ? diag_blocks : 8192)), dim3(256), 0, s, st->mono,
                       st->n, st->dim);
    return sc_check_launch("sc_sample_initial (monodromy blocks)");
}
