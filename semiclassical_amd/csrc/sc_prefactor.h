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
            // one candidate row per lane (d <= 64): |a_ik|^2 of row i = k + lane, DPP maximum (no ds_bpermute)
            const int i = k + lane;
            const bool valid = i < d;
            int bi = wave_pivot_row(valid ? c_abs2(A[i * d + k]) : 0.0, i, valid);
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
__device__ void general_prefactor_matrix(const sc_hk_consts &hk, const double *Mqq, const double *Mqp,
                                         const double *Mpq, const double *Mpp, int ldm, cplx *X, cplx *mat) {
    const int D = hk.dim, dp = hk.dprime, tid = threadIdx.x, nth = blockDim.x;
    const cplx *L1 = (const cplx *)hk.L1, *L2 = (const cplx *)hk.L2;
    const cplx *R1 = (const cplx *)hk.R1, *R2 = (const cplx *)hk.R2;
    for (int pass = 0; pass < 2; ++pass) {
        const double *Ma = pass == 0 ? Mqq : Mpp, *Mb = pass == 0 ? Mqp : Mpq;
        const cplx *Ra = pass == 0 ? R1 : R2, *Rb = pass == 0 ? R2 : R1;
        const cplx fb = pass == 0 ? c_make(0.0, -SC_HBAR) : c_make(0.0, 1.0 / SC_HBAR);
        const cplx *L = pass == 0 ? L1 : L2;
        for (int e = tid; e < D * dp; e += nth) {
            const int a = e / dp, j = e - a * dp;
            cplx s1 = c_make(0, 0), s2 = c_make(0, 0);
            for (int b = 0; b < D; ++b) {
                const double ma = Ma[a * ldm + b], mb = Mb[a * ldm + b];
                const cplx ra = Ra[b * dp + j], rb = Rb[b * dp + j];
                s1.x = fma(ma, ra.x, s1.x); s1.y = fma(ma, ra.y, s1.y);
                s2.x = fma(mb, rb.x, s2.x); s2.y = fma(mb, rb.y, s2.y);
            }
            X[e] = c_add(s1, c_mul(fb, s2));
        }
        __syncthreads();
        for (int e = tid; e < dp * dp; e += nth) {
            const int i = e / dp, j = e - i * dp;
            cplx s = c_make(0, 0);
            for (int a = 0; a < D; ++a) s = c_fma(L[i * D + a], X[a * dp + j], s);
            s = c_scale(s, 0.5);
            mat[e] = pass == 0 ? s : c_add(mat[e], s);
        }
        __syncthreads();
    }
}

}  // namespace
