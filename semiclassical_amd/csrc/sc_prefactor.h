// LDS-resident pieces of the HK prefactor shared by the general step kernels:
// the projected prefactor matrix for dense / rank-deficient width matrices and the pivoted LU determinant.
#pragma once
#include "sc_common.h"

namespace {

// det of the d x d complex matrix A (LDS, row-major), LU with partial pivoting; A is destroyed.
// Every thread returns the determinant.  `ipiv` is one LDS int.
__device__ cplx lds_lu_det(cplx *A, int d, int *ipiv) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    cplx det = c_make(1.0, 0.0);
    for (int k = 0; k < d; ++k) {
        if (wave == 0) {
            // candidate rows i = k + lane (and k + lane + 64 for d > 64): the larger one per lane, then a DPP maximum
            int i = k + lane;
            const bool valid = i < d;
            double mag = valid ? c_abs2(A[i * d + k]) : 0.0;
            if (i + 64 < d) {
                const double m2 = c_abs2(A[(i + 64) * d + k]);
                if (m2 > mag) { mag = m2; i += 64; }
            }
            int bi = wave_pivot_row(mag, i, valid);
            if (bi < 0) bi = k;
            if (bi != k) {
                for (int j = k + lane; j < d; j += 64) {
                    cplx t = A[k * d + j];
                    A[k * d + j] = A[bi * d + j];
                    A[bi * d + j] = t;
                }
            }
            if (lane == 0) *ipiv = bi;
        }
        __syncthreads();
        const cplx piv = A[k * d + k];
        det = c_mul(det, piv);
        if (*ipiv != k) det = c_make(-det.x, -det.y);
        if (piv.x == 0.0 && piv.y == 0.0) {        // singular: uniform exit
            __syncthreads();
            return c_make(0.0, 0.0);
        }
        const cplx inv = c_inv(piv);
        for (int i = k + 1 + wave; i < d; i += nw) {
            const cplx l = c_mul(A[i * d + k], inv);
            for (int j = k + 1 + lane; j < d; j += 64) A[i * d + j] = c_fnma(l, A[k * d + j], A[i * d + j]);
        }
        __syncthreads();
    }
    return det;
}

// General (dense Gamma / rank-deficient) prefactor matrix
//   mat' = 1/2 [ L1 (Mqq R1 - i hbar Mqp R2) + L2 (Mpp R2 + i/hbar Mpq R1) ]       (d' x d')
// M planes are read through generic pointers (global or LDS), leading dimension ldm, plane offsets given.
// Both products are register tiled (3 x 3 outputs per thread): every operand element fetched from L2 / LDS feeds
// three multiply-adds instead of one -- the untiled loops were bound by the load instructions, not the FMAs.
template <int PF_T>
__device__ void general_prefactor_matrix_t(const sc_hk_consts &hk, const double *Mqq, const double *Mqp,
                                           const double *Mpq, const double *Mpp, int ldm, cplx *X, cplx *mat, int panel) {
    // `panel` columns of X = M R at a time (panel = d': all at once); X then holds D x panel values
    const int D = hk.dim, dp = hk.dprime, tid = threadIdx.x, nth = blockDim.x;
    const cplx *L1 = (const cplx *)hk.L1, *L2 = (const cplx *)hk.L2;
    const cplx *R1 = (const cplx *)hk.R1, *R2 = (const cplx *)hk.R2;
    const int ta = (D + PF_T - 1) / PF_T, ti = (dp + PF_T - 1) / PF_T;
    for (int pass = 0; pass < 2; ++pass) {
        const double *Ma = pass == 0 ? Mqq : Mpp, *Mb = pass == 0 ? Mqp : Mpq;
        const cplx *Ra = pass == 0 ? R1 : R2, *Rb = pass == 0 ? R2 : R1;
        const cplx fb = pass == 0 ? c_make(0.0, -SC_HBAR) : c_make(0.0, 1.0 / SC_HBAR);
        const cplx *L = pass == 0 ? L1 : L2;
        for (int jb = 0; jb < dp; jb += panel) {
            const int pw = min(panel, dp - jb), tj = (pw + PF_T - 1) / PF_T;
            // X = Ma Ra + fb Mb Rb   (D x pw)
            for (int t = tid; t < ta * tj; t += nth) {
                const int a0 = (t / tj) * PF_T, j0 = (t % tj) * PF_T;
                cplx s1[PF_T][PF_T], s2[PF_T][PF_T];
#pragma unroll
                for (int u = 0; u < PF_T; ++u)
#pragma unroll
                    for (int v = 0; v < PF_T; ++v) { s1[u][v] = c_make(0, 0); s2[u][v] = c_make(0, 0); }
                for (int b = 0; b < D; ++b) {
                    double ma[PF_T], mb[PF_T];
                    cplx ra[PF_T], rb[PF_T];
#pragma unroll
                    for (int u = 0; u < PF_T; ++u) {
                        const int a = min(a0 + u, D - 1), j = jb + min(j0 + u, pw - 1);
                        ma[u] = Ma[a * ldm + b]; mb[u] = Mb[a * ldm + b];
                        ra[u] = Ra[b * dp + j]; rb[u] = Rb[b * dp + j];
                    }
#pragma unroll
                    for (int u = 0; u < PF_T; ++u)
#pragma unroll
                        for (int v = 0; v < PF_T; ++v) {
                            s1[u][v].x = fma(ma[u], ra[v].x, s1[u][v].x); s1[u][v].y = fma(ma[u], ra[v].y, s1[u][v].y);
                            s2[u][v].x = fma(mb[u], rb[v].x, s2[u][v].x); s2[u][v].y = fma(mb[u], rb[v].y, s2[u][v].y);
                        }
                }
#pragma unroll
                for (int u = 0; u < PF_T; ++u)
#pragma unroll
                    for (int v = 0; v < PF_T; ++v)
                        if (a0 + u < D && j0 + v < pw) X[(a0 + u) * panel + j0 + v] = c_add(s1[u][v], c_mul(fb, s2[u][v]));
            }
            __syncthreads();
            // mat[:, jb .. jb+pw) (+)= 1/2 L X   (d' x pw)
            for (int t = tid; t < ti * tj; t += nth) {
                const int i0 = (t / tj) * PF_T, j0 = (t % tj) * PF_T;
                cplx s[PF_T][PF_T];
#pragma unroll
                for (int u = 0; u < PF_T; ++u)
#pragma unroll
                    for (int v = 0; v < PF_T; ++v) s[u][v] = c_make(0, 0);
                for (int a = 0; a < D; ++a) {
                    cplx l[PF_T], x[PF_T];
#pragma unroll
                    for (int u = 0; u < PF_T; ++u) {
                        l[u] = L[min(i0 + u, dp - 1) * D + a];
                        x[u] = X[a * panel + min(j0 + u, pw - 1)];
                    }
#pragma unroll
                    for (int u = 0; u < PF_T; ++u)
#pragma unroll
                        for (int v = 0; v < PF_T; ++v) s[u][v] = c_fma(l[u], x[v], s[u][v]);
                }
#pragma unroll
                for (int u = 0; u < PF_T; ++u)
#pragma unroll
                    for (int v = 0; v < PF_T; ++v)
                        if (i0 + u < dp && j0 + v < pw) {
                            const int e = (i0 + u) * dp + jb + j0 + v;
                            const cplx h = c_scale(s[u][v], 0.5);
                            mat[e] = pass == 0 ? h : c_add(mat[e], h);
                        }
            }
            __syncthreads();
        }
    }
}

// small matrices keep one output per thread (a 3 x 3 tiling would leave most of the workgroup idle)
// `panel` = number of columns of X (D x panel complex values of LDS) formed at a time; 0 = all d' columns
__device__ void general_prefactor_matrix(const sc_hk_consts &hk, const double *Mqq, const double *Mqp,
                                         const double *Mpq, const double *Mpp, int ldm, cplx *X, cplx *mat,
                                         int panel = 0) {
    if (panel <= 0 || panel > hk.dprime) panel = hk.dprime;
    if (hk.dim * hk.dprime >= 4 * (int)blockDim.x) general_prefactor_matrix_t<3>(hk, Mqq, Mqp, Mpq, Mpp, ldm, X, mat, panel);
    else general_prefactor_matrix_t<1>(hk, Mqq, Mqp, Mpq, Mpp, ldm, X, mat, panel);
}

}  // namespace
