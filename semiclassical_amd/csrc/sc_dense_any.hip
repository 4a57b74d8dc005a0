// Monodromy RK4 and HK prefactor for ANY dimension: the fallback of sc_dense_mono_step beyond D = 96, where neither the
// register-resident MFMA kernels nor an LDS-resident prefactor matrix hold a trajectory any more.  The reference has no
// size limit (propagators.py:313-383, 951-1004); this path keeps that property at the price of speed: every matrix of
// a trajectory lives in a per-workgroup block of global scratch (L2 / MALL resident while it is worked on), products go
// through 16 x 16 LDS tiles on the vector ALUs, the determinant is the pivoted LU of sc_prefactor.h run on global memory
// (the wavefronts of a workgroup share one CU and its write-through L1, so __syncthreads() orders these accesses).
//
// Scratch per workgroup (doubles): stage inputs A0, A1 and the RK4 sum (4 D^2 each), then X1, X2 (D x d' complex each)
// and the d' x d' complex prefactor matrix.
#include "sc_common.h"
#include "sc_prefactor.h"

namespace {

struct AnyArgs {
    sc_state st;
    sc_hk_consts hk;
    const double *inv_mass, *hess;
    int64_t hess_stride, stage_stride;
    double *scratch;
    size_t stride;          // doubles per workgroup
    double dt;
    int mode;
};

// k = f(in):  kqq = W Mpq, kqp = W Mpp, kpq = -H Mqq, kpp = -H Mqp  with  H(a, g) = image[g D + a];  then
//   sum (+)= wgt k   and   out = y0 + c k   (out may be NULL)
__device__ void any_stage(const double *in, const double *y0, const double *image, const double *inv_mass, int D,
                          double wgt, bool first, double c, double *sum, double *out, double (*Ht)[17], double (*Qt)[17]) {
    const int DD = D * D, tid = threadIdx.x, ty = tid >> 4, tx = tid & 15, T = (D + 15) / 16;
    // rows of the q blocks: plain scaling of the p blocks
    for (int e = tid; e < 2 * DD; e += 256) {
        const int blk = e / DD, ab = e - blk * DD, a = ab / D;
        const double k = in[(2 + blk) * DD + ab] * inv_mass[a];
        sum[blk * DD + ab] = first ? wgt * k : sum[blk * DD + ab] + wgt * k;
        if (out) out[blk * DD + ab] = y0[blk * DD + ab] + c * k;
    }
    // p blocks: -H Q for Q = Mqq (-> kpq) and Q = Mqp (-> kpp), 16 x 16 output tiles
    for (int blk = 0; blk < 2; ++blk) {
        const double *Q = in + blk * DD;
        for (int tile = 0; tile < T * T; ++tile) {
            const int a = 16 * (tile / T) + ty, b = 16 * (tile % T) + tx;
            double acc = 0.0;
            for (int g0 = 0; g0 < D; g0 += 16) {
                __syncthreads();
                const int ga = g0 + tx, gq = g0 + ty;
                Ht[ty][tx] = (a < D && ga < D) ? image[(size_t)ga * D + a] : 0.0;        // H(a, g0 + tx)
                Qt[ty][tx] = (gq < D && b < D) ? Q[(size_t)gq * D + b] : 0.0;            // Q(g0 + ty, b)
                __syncthreads();
#pragma unroll
                for (int g = 0; g < 16; ++g) acc = fma(Ht[ty][g], Qt[g][tx], acc);
            }
            if (a < D && b < D) {
                const int e = (2 + blk) * DD + a * D + b;
                const double k = -acc;
                sum[e] = first ? wgt * k : sum[e] + wgt * k;
                if (out) out[e] = y0[e] + c * k;
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void mono_rk4_any_kernel(AnyArgs A) {
    __shared__ double Ht[16][17], Qt[16][17];
    const int D = A.st.dim, DD = D * D, tid = threadIdx.x;
    double *A0 = A.scratch + (size_t)blockIdx.x * A.stride, *A1 = A0 + 4 * (size_t)DD, *SUM = A1 + 4 * (size_t)DD;
    const double dt = A.dt, hh = 0.5 * dt, h6 = dt / 6.0;
    for (int64_t tr = blockIdx.x; tr < A.st.n; tr += gridDim.x) {
        double *M = A.st.mono + tr * 4 * (int64_t)DD;
        const double *H = A.hess + tr * A.hess_stride;
        any_stage(M, M, H, A.inv_mass, D, 1.0, true, hh, SUM, A0, Ht, Qt);
        any_stage(A0, M, H + A.stage_stride, A.inv_mass, D, 2.0, false, hh, SUM, A1, Ht, Qt);
        any_stage(A1, M, H + 2 * A.stage_stride, A.inv_mass, D, 2.0, false, dt, SUM, A0, Ht, Qt);
        any_stage(A0, M, H + 3 * A.stage_stride, A.inv_mass, D, 1.0, false, 0.0, SUM, nullptr, Ht, Qt);
        for (int e = tid; e < 4 * DD; e += 256) M[e] += h6 * SUM[e];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void prefactor_any_kernel(AnyArgs A) {
    __shared__ int ipiv;
    const int D = A.st.dim, DD = D * D, dp = A.hk.dprime, tid = threadIdx.x;
    double *base = A.scratch + (size_t)blockIdx.x * A.stride + 12 * (size_t)DD;
    cplx *X1 = (cplx *)base, *X2 = X1 + (size_t)D * dp, *mat = X2 + (size_t)D * dp;
    const cplx *L1 = (const cplx *)A.hk.L1, *L2 = (const cplx *)A.hk.L2, *R1 = (const cplx *)A.hk.R1, *R2 = (const cplx *)A.hk.R2;
    for (int64_t tr = blockIdx.x; tr < A.st.n; tr += gridDim.x) {
        const double *M = A.st.mono + tr * 4 * (int64_t)DD;
        __syncthreads();
        if (A.hk.diag) {
            for (int e = tid; e < DD; e += 256) {
                const int a = e / D, b = e - a * D;
                const double sta = A.hk.st[a], sib = A.hk.si[b];
                mat[e] = c_make(0.5 * (sta / sib * M[e] + sib / sta * M[3 * DD + e]),
                                0.5 * (-SC_HBAR * sta * sib * M[DD + e] + M[2 * DD + e] / (SC_HBAR * sta * sib)));
            }
        } else {
            // X1 = Mqq R1 - i hbar Mqp R2,  X2 = Mpp R2 + i/hbar Mpq R1  (D x d'),  mat = 1/2 (L1 X1 + L2 X2)   (:969-994)
            for (int e = tid; e < D * dp; e += 256) {
                const int a = e / dp, j = e - a * dp;
                cplx s1 = c_make(0, 0), s2 = c_make(0, 0), t1 = c_make(0, 0), t2 = c_make(0, 0);
                for (int b = 0; b < D; ++b) {
                    const cplx r1 = R1[b * dp + j], r2 = R2[b * dp + j];
                    const double qq = M[a * D + b], qp = M[DD + a * D + b], pq = M[2 * DD + a * D + b], pp = M[3 * DD + a * D + b];
                    s1.x = fma(qq, r1.x, s1.x); s1.y = fma(qq, r1.y, s1.y); s2.x = fma(qp, r2.x, s2.x); s2.y = fma(qp, r2.y, s2.y);
                    t1.x = fma(pp, r2.x, t1.x); t1.y = fma(pp, r2.y, t1.y); t2.x = fma(pq, r1.x, t2.x); t2.y = fma(pq, r1.y, t2.y);
                }
                X1[e] = c_add(s1, c_mul(c_make(0.0, -SC_HBAR), s2));
                X2[e] = c_add(t1, c_mul(c_make(0.0, 1.0 / SC_HBAR), t2));
            }
            __syncthreads();
            for (int e = tid; e < dp * dp; e += 256) {
                const int i = e / dp, j = e - i * dp;
                cplx s = c_make(0, 0);
                for (int a = 0; a < D; ++a) {
                    s = c_fma(L1[i * D + a], X1[a * dp + j], s);
                    s = c_fma(L2[i * D + a], X2[a * dp + j], s);
                }
                mat[e] = c_scale(s, 0.5);
            }
        }
        __syncthreads();
        const cplx det = lds_lu_det(mat, dp, &ipiv);       // pointer-generic: here on global memory
        if (tid == 0) {
            cplx *c2 = (cplx *)A.st.c2;
            if (A.mode == 0) {
                const cplx prev = c2[tr];
                if (prev.x < 0.0 && det.x < 0.0 && prev.y * det.y < 0.0) A.st.sgn[tr] = -A.st.sgn[tr];
            } else {
                A.st.sgn[tr] = 1.0;
            }
            c2[tr] = det;
        }
    }
}

}  // namespace

static int any_grid(int64_t n) { return (int)(n < 256 ? (n > 0 ? n : 1) : 256); }
static size_t any_stride(int D, int dp) { return 12 * (size_t)D * D + 2 * (2 * (size_t)D * dp + (size_t)dp * dp); }

// bytes of scratch sc_dense_mono_step needs in `mono_sums` for n trajectories of dimension D (d' = rank of the widths)
extern "C" int64_t sc_dense_mono_scratch_bytes(int64_t n, int32_t D, int32_t dprime) {
    if (D < 1 || dprime < 1 || dprime > D || n < 0) return -1;
    if (D <= 64) return 0;
    if (D <= 96) return (int64_t)n * 4 * D * D * (int64_t)sizeof(double);          // RK4 sums of the MFMA kernel
    return (int64_t)any_grid(n) * (int64_t)any_stride(D, dprime) * (int64_t)sizeof(double);
}

int sc_launch_dense_any(const sc_state *st, const sc_hk_consts *hk, const double *inv_mass, const double *hess,
                        double *scratch, double dt, int mode, hipStream_t s) {
    const int D = st->dim;
    AnyArgs a{*st, *hk, inv_mass, hess, 4 * (int64_t)D * D, (int64_t)D * D, scratch, any_stride(D, hk->dprime), dt, mode};
    const int grid = any_grid(st->n);
    if (mode == 0) {
        hipLaunchKernelGGL(mono_rk4_any_kernel, dim3(grid), dim3(256), 0, s, a);
        const int rc = sc_check_launch("sc_dense_mono_step (any-D RK4)");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(prefactor_any_kernel, dim3(grid), dim3(256), 0, s, a);
    return sc_check_launch("sc_dense_mono_step (any-D prefactor)");
}
