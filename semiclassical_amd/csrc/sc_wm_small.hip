// Walton-Manolopoulos prefactor and correlation terms for SMALL matrices (D <= 16, e = 2 d' <= 16): register-resident.
//
// The same arithmetic as wm_kernel in sc_wm.hip (reference semiclassical/propagators.py:1132-1389 _expand_L /
// _prefactor, :1577-1614 eqn (85), :1652-1719 eqn (100)) with a different mapping to the machine:
//
//   * one trajectory per 16-lane DPP row, four trajectories per wavefront, sixteen per workgroup; lane r of a row holds
//     ROW r of every matrix of its trajectory in registers (D, d' are template parameters: all loops are unrolled and
//     every register index is static);
//   * EVERY matrix product is a chain of v_fmac_f64_dpp row_newbcast (sc_row16.h: the 16-lane broadcast of the other
//     factor's row fused into the multiply-add, one instruction per multiply-add, no cross-lane move, no LDS, no scalar
//     operand traffic): C[r][j] += A[r][k] * B[k][j] takes B[k][j] from lane k, and products with a constant matrix take
//     the constant's row k from lane k as well (the per-lane rows of the constants are staged in LDS once per workgroup
//     and read where they are used);
//   * the two linear systems are eliminated IN A FIXED PIVOT ORDER, so that the pivot row is a static lane and the
//     elimination is made of the same fused instructions (round 2 searched the pivot among the lanes and fetched its row
//     with ds_bpermute: 1100 LDS operations per four trajectories on the critical path).  W A' = BQ' (eqns 57, 59 need
//     W = BQ' A'^-1) is solved by column operations on [A'; BQ'] with lane j holding row j of both -- forward elimination
//     with normalised pivot rows, then back substitution -- which leaves row a of W in lane a, where the following
//     products want it (no transposition).  M' rho = hat is solved for its five right-hand sides by row operations.
//     A pivot is accepted if it is within a factor 16 of the largest candidate partial pivoting could have chosen
//     (threshold pivoting; candidates are registers of the pivot lane resp. lanes of the pivot column); otherwise the
//     trajectory is flagged (sc_wm_consts.flags) and recomputed by the fully pivoted LDS kernel wm_kernel<false> in the
//     same stream -- the scheme of the HK fast path (sc_hk_step_sd.hip);
//   * LDS holds the per-lane rows of the constants and one D x (e|1) transposition buffer per trajectory (columns of Mq',
//     Mp'); there is no __syncthreads in the loop;  the kernel is compiled for two wavefronts per SIMD (256 registers).
//
// Restatement (see sc_wm.hip for the derivation): e = 2d', Mq' = [Mqq U, Mqp U], Mp' = [Mpq U, Mpp U],
//   A' = Cst' + Mq'^T Gt Mq' + i/hbar (2G - H),  G = Mp'^T Mq',  H[i][j] = i < d' ? G[i][j] : G[j][i],
//   BQ' = Gt Mq' + i/hbar Mp',  Wm = BQ' A'^-1,  Gt~ = Gt - Wm BQ'^T (57),  Gti = Wm Bq'^T (59),
//   V = Gti iGi0,  CQQ = Gt~ - V Gti^T (70),  M' = U^T (G0 + CQQ) U,  rho_v = M'^-1 U^T v.
#include "sc_wm.h"
#include "sc_row16.h"

namespace {

// bytes of LDS: per-lane rows of the constants (doubles) + per-trajectory transposition buffers (doubles)
template <int D, int DP>
struct WmSmallLayout {
    static constexpr int E = 2 * DP, EP = E | 1;
    // row pitches of the per-lane constants: lane r reads ITS row, so the pitch decides the banks the 16 lanes of a row meet
    // on (rows 8 D bytes apart: lanes r and r + 8 on one bank at D = 12; measured SQ_LDS_BANK_CONFLICT = 61 % of the LDS cycles).
    // One element more per row spreads them (as in sc_hk_step_lin.hip).
    static constexpr int PD = D + 1, PDP = DP + 1, PE = E + 1;
    static constexpr int n_const = 5 * 16 * PD       // rows of Gt, G0, Cqq, S, iGi0, zero padded to 16 lanes
                                   + 16 * PD         // UT[i][a] = U[a][i], zero for i >= d'
                                   + 3 * 16 * PDP    // rows of U, Re Bq'[:, :d'] = Gamma_i U, Im Bq'[:, d':] = -U / hbar
                                   + (3 * 16 * PDP & 1)   // keeps the complex rows below 16-byte aligned
                                   + 2 * 16 * PE     // rows of Cst' / s (complex), zero for r >= e
                                   + 8 * 16;         // q0, p0, n1, s_n1, w_n1, crow = Cqq n1, 2 spare
    static constexpr int PB = 2 * E + 2;             // doubles per lane of the parked BQ' row (pitch: 16-byte aligned, odd in 16-byte units)
    static constexpr int xt = D * EP + 16;           // transposition buffer (+16: lanes beyond the matrix read, never use)
    static constexpr int xbuf = (xt > 16 * PB ? xt : 16 * PB) + (xt > 16 * PB ? xt & 1 : 0);   // doubles per trajectory
    static constexpr size_t bytes = (size_t)n_const * 8 + (size_t)16 * xbuf * 8;
};

#ifndef SC_WM_SMALL_OCC
// waves per SIMD the kernel is compiled for (register budget 512 / OCC)
#define SC_WM_SMALL_OCC 2
#endif
#ifndef SC_WM_FORCE_WEAK
#define SC_WM_FORCE_WEAK 0     // 1: variant library that flags EVERY trajectory (tests of the pivoted fallback)
#endif

// sum over the lanes k < N of the 16-lane row of every element of t (result in all lanes): t_m <- sum_k t_m[k]
template <int N, int M>
__device__ __forceinline__ void sum_rows(double (&t)[M], const double &one) {
    double s[M];
#pragma unroll
    for (int m = 0; m < M; ++m) s[m] = 0.0;
    dpp_guard(t);
    sfor<0, N>([&](auto kc) {
#pragma unroll
        for (int m = 0; m < M; ++m) fmac_bc<decltype(kc)::value>(s[m], t[m], one);
    });
#pragma unroll
    for (int m = 0; m < M; ++m) t[m] = s[m];
}
template <int N, int M>
__device__ __forceinline__ void sum_rows(cplx (&t)[M], const double &one) {
    double s[2 * M];
#pragma unroll
    for (int m = 0; m < M; ++m) { s[2 * m] = t[m].x; s[2 * m + 1] = t[m].y; }
    sum_rows<N>(s, one);
#pragma unroll
    for (int m = 0; m < M; ++m) t[m] = c_make(s[2 * m], s[2 * m + 1]);
}


template <int D, int DP>
__global__ __launch_bounds__(256, SC_WM_SMALL_OCC) void wm_small_kernel(WmArgs A) {
    typedef WmSmallLayout<D, DP> L;
    constexpr int E = L::E, EP = L::EP, DD = D * D;
    extern __shared__ double2 smem2[];
    const int tid = threadIdx.x, r = tid & 15, grp = tid >> 4;
    const sc_wm_consts &W = A.wc;
    const double ihb = 1.0 / SC_HBAR, sA = W.inv_scale_a;

    double *ls = (double *)smem2;
    constexpr int PD = L::PD, PDP = L::PDP, PE = L::PE;
    double *sGt = ls;   ls += 16 * PD;
    double *sG0 = ls;   ls += 16 * PD;
    double *sCqq = ls;  ls += 16 * PD;
    double *sS = ls;    ls += 16 * PD;
    double *siG = ls;   ls += 16 * PD;
    double *sUT = ls;   ls += 16 * PD;
    double *sU = ls;    ls += 16 * PDP;
    double *sBr = ls;   ls += 16 * PDP;
    double *sBi = ls;   ls += 16 * PDP + (3 * 16 * PDP & 1);
    double *sCst = ls;  ls += 2 * 16 * PE;
    double *cvec = ls;  ls += 8 * 16;            // [q0 | p0 | n1 | s_n1 | w_n1 | crow | - | -][16]
    double *xall = ls;

    // ---- stage the per-lane rows of the constants (once per workgroup) ----
    for (int e = tid; e < 16 * D; e += 256) {
        const int i = e / D, b = e - i * D;
        const bool in = i < D;
        const int o = i * PD + b;
        sGt[o] = in ? W.Gt[i * D + b] : 0.0;
        sG0[o] = in ? W.G0[i * D + b] : 0.0;
        sCqq[o] = in ? W.Cqq[i * D + b] : 0.0;
        sS[o] = in ? W.S[i * D + b] : 0.0;
        siG[o] = in ? W.iGi0[i * D + b] : 0.0;
        sUT[o] = i < DP ? W.U[b * DP + i] : 0.0;
    }
    for (int e = tid; e < 16 * DP; e += 256) {
        const int i = e / DP, k = e - i * DP;
        const bool in = i < D;
        const int o = i * PDP + k;
        sU[o] = in ? W.U[i * DP + k] : 0.0;
        sBr[o] = in ? W.Bq[2 * (i * E + k)] : 0.0;                   // Re Bq'[i][k]      = (Gamma_i U)[i][k]
        sBi[o] = in ? W.Bq[2 * (i * E + DP + k) + 1] : 0.0;          // Im Bq'[i][d' + k] = -U[i][k] / hbar
    }
    for (int e = tid; e < 16 * E; e += 256) {
        const int i = e / E, j = e - i * E;
        sCst[2 * (i * PE + j)] = i < E ? W.Cst[2 * (i * E + j)] * sA : 0.0;
        sCst[2 * (i * PE + j) + 1] = i < E ? W.Cst[2 * (i * E + j) + 1] * sA : 0.0;
    }
    if (tid < 16) {
        const bool in = tid < D, nac = in && A.has_nac;
        cvec[tid] = in ? W.q0[tid] : 0.0;
        cvec[16 + tid] = in ? W.p0[tid] : 0.0;
        cvec[32 + tid] = nac ? W.n1[tid] : 0.0;
        cvec[48 + tid] = nac ? W.s_n1[tid] : 0.0;
        cvec[64 + tid] = nac ? W.w_n1[tid] : 0.0;
        double c = 0.0;
        if (nac) for (int b = 0; b < D; ++b) c = fma(W.Cqq[tid * D + b], W.n1[b], c);
        cvec[80 + tid] = c;
    }
    __syncthreads();

    double *xr = xall + grp * L::xbuf;          // transposition buffer of this trajectory
    const int64_t n = A.st.n, stride = (int64_t)gridDim.x * 16;
    // The passes run from the LAST trajectories to the first: the step kernel in front of this launch wrote the monodromy
    // blocks in ascending order, so the highest ones are what the memory-side cache (256 MB of a 460 MB state at n = 1e5)
    // still holds; and the next step kernel, ascending again, starts on what this launch read last.
    const int64_t first_t0 = (int64_t)blockIdx.x * 16;
    const int64_t npass = first_t0 < n ? (n - first_t0 + stride - 1) / stride : 0;
    for (int64_t pass = npass - 1; pass >= 0; --pass) {
        const int64_t t0 = first_t0 + pass * stride;
        const bool active = t0 + grp < n;
        const int64_t tr = active ? t0 + grp : n - 1;      // idle rows redo the last trajectory and discard it
        const double *qp = A.st.qp + tr * 2 * D, *zi = A.zi + tr * 2 * D;
        const double *M = A.st.mono + tr * 4 * (int64_t)DD;
        // the per-lane rows of the constants are read from LDS where they are used, never kept across phases
        int lofs = 0;
        asm volatile("" : "+v"(lofs));
        const double *cGt = sGt + lofs + r * PD, *cG0 = sG0 + lofs + r * PD, *cCqq = sCqq + lofs + r * PD, *cS = sS + lofs + r * PD;
        const double *ciG = siG + lofs + r * PD, *cUT = sUT + lofs + r * PD, *cU = sU + lofs + r * PDP, *cBr = sBr + lofs + r * PDP;
        const double *cBi = sBi + lofs + r * PDP;
        const cplx *cCst = (const cplx *)(sCst + lofs) + r * PE;
        const double *cv = cvec + lofs;

        // ---- rows of the monodromy blocks; Mq' = [Mqq U, Mqp U], Mp' = [Mpq U, Mpp U] (row r) ----
        double Mq[E], Mp[E];
        double qv = 0.0, pv = 0.0, dq = 0.0, dpv = 0.0, dQ = 0.0;
        {
            double m0[D], m1[D], m2[D], m3[D];
#pragma unroll
            for (int b = 0; b < D; ++b) { m0[b] = 0.0; m1[b] = 0.0; m2[b] = 0.0; m3[b] = 0.0; }
            if (r < D) {
#pragma unroll
                for (int b = 0; b < D; ++b) {
                    m0[b] = M[r * D + b]; m1[b] = M[DD + r * D + b];
                    m2[b] = M[2 * DD + r * D + b]; m3[b] = M[3 * DD + r * D + b];
                }
                qv = qp[r]; pv = qp[D + r];
                dq = cv[r] - zi[r]; dpv = cv[16 + r] - zi[D + r];
                dQ = cv[r] - qv;
            }
            double ur[DP];
#pragma unroll
            for (int j = 0; j < DP; ++j) ur[j] = cU[j];
#pragma unroll
            for (int j = 0; j < E; ++j) { Mq[j] = 0.0; Mp[j] = 0.0; }
            dpp_guard(ur);
            sfor<0, D>([&](auto bcn) {
                constexpr int b = decltype(bcn)::value;
#pragma unroll
                for (int j = 0; j < DP; ++j) {
                    fmac_bc<b>(Mq[j], ur[j], m0[b]); fmac_bc<b>(Mq[DP + j], ur[j], m1[b]);
                    fmac_bc<b>(Mp[j], ur[j], m2[b]); fmac_bc<b>(Mp[DP + j], ur[j], m3[b]);
                }
            });
        }

        // ---- columns of Mq', Mp' (lane i < e holds column i) through the transposition buffer ----
        double MqT[D], MpT[D];
        {
            const int rc = r < E ? r : E - 1;
            if (r < D) {
#pragma unroll
                for (int j = 0; j < E; ++j) xr[r * EP + j] = Mq[j];
            }
            wave_lds_fence();
#pragma unroll
            for (int a = 0; a < D; ++a) { const double x = xr[a * EP + rc]; MqT[a] = r < E ? x : 0.0; }
            wave_lds_fence();
            if (r < D) {
#pragma unroll
                for (int j = 0; j < E; ++j) xr[r * EP + j] = Mp[j];
            }
            wave_lds_fence();
#pragma unroll
            for (int a = 0; a < D; ++a) { const double x = xr[a * EP + rc]; MpT[a] = r < E ? x : 0.0; }
            wave_lds_fence();
        }
        // ---- Tq = Gt Mq' (row r) ----
        double Tq[E];
        {
            double gt[D];
#pragma unroll
            for (int b = 0; b < D; ++b) gt[b] = cGt[b];
#pragma unroll
            for (int j = 0; j < E; ++j) Tq[j] = 0.0;
            dpp_guard(Mq);
            sfor<0, D>([&](auto bcn) {
                constexpr int b = decltype(bcn)::value;
#pragma unroll
                for (int j = 0; j < E; ++j) fmac_bc<b>(Tq[j], Mq[j], gt[b]);
            });
        }

        // ---- row j = r of A'/s:  Re = Cst + Mq'^T Tq,  Im = Cst + i/hbar (2G - H):  H[j][i] = j < d' ? G[j][i] : G[i][j],
        //      G[j][i] = sum_a Mp'[a][j] Mq'[a][i] -- accumulated straight into the matrix, coefficients folded per lane ----
        cplx rowA[E];
        {
#pragma unroll
            for (int i = 0; i < E; ++i) rowA[i] = cCst[i];
            const double sel = r < DP ? 0.0 : 1.0;
            const double fa = (1.0 + sel) * ihb * sA, fb = -sel * ihb * sA;
            double cs[D], ca[D], cb[D];
#pragma unroll
            for (int a = 0; a < D; ++a) { cs[a] = MqT[a] * sA; ca[a] = MpT[a] * fa; cb[a] = MqT[a] * fb; }
            dpp_guard(Tq, Mq, Mp);
            sfor<0, D>([&](auto ac) {
                constexpr int a = decltype(ac)::value;
#pragma unroll
                for (int i = 0; i < E; ++i) {
                    fmac_bc<a>(rowA[i].x, Tq[i], cs[a]);
                    fmac_bc<a>(rowA[i].y, Mq[i], ca[a]);
                    fmac_bc<a>(rowA[i].y, Mp[i], cb[a]);
                }
            });
        }
        // BQ' (row r) = Tq + i/hbar Mp': the elimination works on it in place; a copy for eqn (57) waits in LDS
        cplx rowB[E];
        cplx *park = (cplx *)(xr + r * L::PB);
#pragma unroll
        for (int k = 0; k < E; ++k) { rowB[k] = c_make(Tq[k], ihb * Mp[k]); park[k] = rowB[k]; }

        // ---- W (A'/s) = BQ' by column operations in the fixed pivot order 0 .. e-1 (see the header) ----
        cplx detA = c_make(1.0, 0.0);
        int weak = SC_WM_FORCE_WEAK;
        dpp_guard(rowA, rowB);
        sfor<0, E>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            // candidates of partial pivoting: entries k .. e-1 of the pivot lane's row
            int key_max = 0;
            sfor<k + 1, E>([&](auto ic) { key_max = max(key_max, __double2hiint(c_abs2(rowA[decltype(ic)::value]))); });
            const double mag = c_abs2(rowA[k]);
            if (r == k && (weak_pivot_keys(__double2hiint(mag), key_max) || mag == 0.0)) weak = 1;
            const cplx piv = c_make(bc<k>(rowA[k].x), bc<k>(rowA[k].y));
            detA = c_mul(detA, piv);
            const cplx inv = c_inv_newton(piv);
            const cplx tB = c_mul(rowB[k], inv);
            const cplx tAf = c_mul(rowA[k], inv);
            rowA[k] = tAf; rowB[k] = tB;
            // the pivot lane's own entries below the diagonal are never read again: its multiplier is zeroed so that
            // the in-place update below does not change what the other lanes read from it
            const cplx tA = c_make(r == k ? 0.0 : tAf.x, r == k ? 0.0 : tAf.y);
            dpp_guard(rowA, rowB);
            sfor<k + 1, E>([&](auto ic) { cfnma_bc<k>(rowB[decltype(ic)::value], rowA[decltype(ic)::value], tB); });
            cfnma_inplace_range<k, k + 1, E>(rowA, tA);
        });
        sfor<1, E>([&](auto kc) {                      // back substitution, column k of the unit upper triangle
            constexpr int k = E - decltype(kc)::value;
            sfor<0, k>([&](auto ic) { cfnma_bc<k>(rowB[decltype(ic)::value], rowA[decltype(ic)::value], rowB[k]); });
        });

        // ---- Wm = BQ' A'^-1 (row r);  Gt~ = Gt - Wm BQ'^T (57) ----
        cplx Gtl[D], Gti[D];
        {
            cplx Wm[E], BQ[E];
#pragma unroll
            for (int k = 0; k < E; ++k) { Wm[k] = c_scale(rowB[k], sA); BQ[k] = park[k]; }
#pragma unroll
            for (int b = 0; b < D; ++b) Gtl[b] = c_make(cGt[b], 0.0);
            dpp_guard(BQ, Gtl);
            sfor<0, D>([&](auto bcn) {
                constexpr int b = decltype(bcn)::value;
#pragma unroll
                for (int k = 0; k < E; ++k) cfnma_bc<b>(Gtl[b], BQ[k], Wm[k]);
            });
            // ---- Gti = Wm Bq'^T (59);  Bq' = [Gamma_i U | -i/hbar U]: real in its first d' columns, imaginary in the last d'
            double br[DP], bi[DP];
#pragma unroll
            for (int k = 0; k < DP; ++k) { br[k] = cBr[k]; bi[k] = cBi[k]; }
#pragma unroll
            for (int b = 0; b < D; ++b) Gti[b] = c_make(0.0, 0.0);
            dpp_guard(br, bi);
            sfor<0, D>([&](auto bcn) {
                constexpr int b = decltype(bcn)::value;
#pragma unroll
                for (int k = 0; k < DP; ++k) {
                    fmac_bc<b>(Gti[b].x, br[k], Wm[k].x); fmac_bc<b>(Gti[b].y, br[k], Wm[k].y);
                    fnmac_bc<b>(Gti[b].x, bi[k], Wm[DP + k].y); fmac_bc<b>(Gti[b].y, bi[k], Wm[DP + k].x);
                }
            });
        }

        // ---- per-trajectory vectors: g = iGi0 (p0 - p_i), s_dq = S dq, w_dQ = G0 dQ, cdq = Cqq dq, g0g = G0 g ----
        double gv = 0.0, sdq = 0.0, wdQ = 0.0, cdq = 0.0, g0g = 0.0;
        const double p0r = cv[16 + r], n1r = cv[32 + r], sn1r = cv[48 + r], wn1r = cv[64 + r], crowr = cv[80 + r];
        cplx y = c_make(r < D ? pv - p0r : 0.0, 0.0), u1 = c_make(0.0, 0.0), u2 = c_make(0.0, 0.0);
        cplx V[D];
        {
            double ig[D];
#pragma unroll
            for (int b = 0; b < D; ++b) ig[b] = ciG[b];
            {
                double sr[D], cq[D], g0[D];
#pragma unroll
                for (int b = 0; b < D; ++b) { sr[b] = cS[b]; cq[b] = cCqq[b]; g0[b] = cG0[b]; }
                dpp_guard(dpv, dq, dQ);
                sfor<0, D>([&](auto bcn) {
                    constexpr int b = decltype(bcn)::value;
                    fmac_bc<b>(gv, dpv, ig[b]); fmac_bc<b>(sdq, dq, sr[b]); fmac_bc<b>(cdq, dq, cq[b]); fmac_bc<b>(wdQ, dQ, g0[b]);
                });
                double sn1 = sn1r;
                dpp_guard(gv, sdq, sn1);
                sfor<0, D>([&](auto bcn) {
                    constexpr int b = decltype(bcn)::value;
                    fmac_bc<b>(g0g, gv, g0[b]);
                    fmac_bc<b>(y.x, gv, Gti[b].x); fmac_bc<b>(y.y, gv, Gti[b].y);
                    fmac_bc<b>(u1.x, sdq, Gti[b].x); fmac_bc<b>(u1.y, sdq, Gti[b].y);
                    fmac_bc<b>(u2.x, sn1, Gti[b].x); fmac_bc<b>(u2.y, sn1, Gti[b].y);
                });
            }
            // ---- V = Gti iGi0 ; CQQ = Gt~ - V Gti^T (70), in place in Gtl ----
#pragma unroll
            for (int b = 0; b < D; ++b) V[b] = c_make(0.0, 0.0);
            dpp_guard(ig);
            sfor<0, D>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
#pragma unroll
                for (int b = 0; b < D; ++b) { fmac_bc<k>(V[b].x, ig[b], Gti[k].x); fmac_bc<k>(V[b].y, ig[b], Gti[k].y); }
            });
        }
        dpp_guard(Gti);
        sfor<0, D>([&](auto bcn) {
            constexpr int b = decltype(bcn)::value;
#pragma unroll
            for (int k = 0; k < D; ++k) cfnma_bc<b>(Gtl[b], Gti[k], V[k]);
        });
        if (W.cqq_out && active && r < D) {
            cplx *out = (cplx *)W.cqq_out + tr * (int64_t)DD + r * D;
#pragma unroll
            for (int b = 0; b < D; ++b) out[b] = Gtl[b];
        }
        if (W.dvec_out && active && r < D)       // C_qQ^T (q0 - q) + i/hbar PI_Q = u_dq + i/hbar (y + p0)
            ((cplx *)W.dvec_out)[tr * (int64_t)D + r] = c_make(u1.x - ihb * y.y, u1.y + ihb * (y.x + p0r));

        // ---- M'/(2 pi) = U^T (G0 + CQQ) U / (2 pi) (row i < d') and hat_v = U^T {u_dq, u_n1, w_dQ, w_n1, y} ----
        cplx Mr[DP], hat[5];
        {
            cplx R[DP];
            double ur[DP], gre[D];
#pragma unroll
            for (int j = 0; j < DP; ++j) { ur[j] = cU[j]; R[j] = c_make(0.0, 0.0); }
#pragma unroll
            for (int b = 0; b < D; ++b) gre[b] = cG0[b] + Gtl[b].x;
            dpp_guard(ur);
            sfor<0, D>([&](auto bcn) {
                constexpr int b = decltype(bcn)::value;
#pragma unroll
                for (int j = 0; j < DP; ++j) { fmac_bc<b>(R[j].x, ur[j], gre[b]); fmac_bc<b>(R[j].y, ur[j], Gtl[b].y); }
            });
            double ut[D];
#pragma unroll
            for (int a = 0; a < D; ++a) ut[a] = cUT[a];
#pragma unroll
            for (int j = 0; j < DP; ++j) Mr[j] = c_make(0.0, 0.0);
#pragma unroll
            for (int v = 0; v < 5; ++v) hat[v] = c_make(0.0, 0.0);
            double wn1 = wn1r;
            dpp_guard(R, u1, u2, wdQ, wn1, y);
            sfor<0, D>([&](auto ac) {
                constexpr int a = decltype(ac)::value;
#pragma unroll
                for (int j = 0; j < DP; ++j) { fmac_bc<a>(Mr[j].x, R[j].x, ut[a]); fmac_bc<a>(Mr[j].y, R[j].y, ut[a]); }
                fmac_bc<a>(hat[0].x, u1.x, ut[a]); fmac_bc<a>(hat[0].y, u1.y, ut[a]);
                fmac_bc<a>(hat[1].x, u2.x, ut[a]); fmac_bc<a>(hat[1].y, u2.y, ut[a]);
                fmac_bc<a>(hat[2].x, wdQ, ut[a]);
                fmac_bc<a>(hat[3].x, wn1, ut[a]);
                fmac_bc<a>(hat[4].x, y.x, ut[a]); fmac_bc<a>(hat[4].y, y.y, ut[a]);
            });
        }
#pragma unroll
        for (int j = 0; j < DP; ++j) Mr[j] = c_scale(Mr[j], W.inv_two_pi);
        // ---- M' rho = hat by row operations in the fixed pivot order (rows >= d' are zero and take no part); the matrix
        //      and its five right-hand sides are one array: ms = [M'/(2 pi) | hat] (row r) ----
        cplx ms[DP + 5];
#pragma unroll
        for (int j = 0; j < DP; ++j) ms[j] = Mr[j];
#pragma unroll
        for (int v = 0; v < 5; ++v) ms[DP + v] = hat[v];
        cplx detM = c_make(1.0, 0.0), myinv = c_make(0.0, 0.0);
        sfor<0, DP>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const double mag = c_abs2(ms[k]);
            const int key = (r >= k && r < DP) ? __double2hiint(mag) : 0;
            const int key_max = row_max(key), key_piv = bc_i32<k>(key);
            if (weak_pivot_keys(key_piv, key_max) || key_piv == 0) weak = 1;      // (a zero or denormal pivot goes to the pivoted kernel)
            const cplx piv = c_make(bc<k>(ms[k].x), bc<k>(ms[k].y));
            detM = c_mul(detM, piv);
            const cplx inv = c_inv_newton(piv);
            const cplx f = c_mul(ms[k], inv);
            const cplx m = c_make(r == k ? 0.0 : f.x, r == k ? 0.0 : f.y);
            if (r == k) myinv = inv;
            dpp_guard(ms);
            cfnma_inplace_range<k, k + 1, DP + 5>(ms, m);
        });
        // lane k holds rho_v[k];  rho = M'^-1 hat (the 1/2pi of the scaling)
        cplx rho[5];
#pragma unroll
        for (int v = 0; v < 5; ++v) rho[v] = c_scale(c_mul(ms[DP + v], myinv), W.inv_two_pi);

        // ---- bilinear forms a^T iM b and the scalar sums over the modes: the per-lane terms first, then all sums over the
        //      lanes together, each as a chain of fused broadcast multiply-adds with 1.0 (lanes k < d' resp. k < D only: no
        //      masking, and a third of the instructions of the rotate-and-add reduction) ----
        enum { UDQ = 0, UN1 = 1, WDQ = 2, WN1 = 3, Y = 4 };
        double one = 1.0;
        asm volatile("" : "+v"(one));
        const double piq = p0r - g0g;                                            // (72)
        cplx f6[6] = {c_mul(hat[Y], rho[Y]), c_mul(hat[UDQ], rho[UDQ]), c_mul(hat[WDQ], rho[WDQ]),
                      c_mul(hat[UDQ], rho[WDQ]), c_mul(hat[UDQ], rho[Y]), c_mul(hat[WDQ], rho[Y])};
        double s8[8] = {dq * cdq, dq * crowr, dQ * wdQ, dQ * wn1r, piq * dq, piq * n1r, p0r * dQ, dpv * gv};
        sum_rows<DP>(f6, one);
        sum_rows<D>(s8, one);
        const double dqCdq = s8[0], dqCn1 = s8[1], dQGdQ = s8[2], dQGn1 = s8[3], piq_dq = s8[4], piq_n1 = s8[5], p0_dQ = s8[6];
        const double eps = -0.5 * ihb * ihb * s8[7];                             // (74), b0 = 0
        const cplx yy = f6[0];
        const cplx gamma = c_make(eps - 0.5 * ihb * ihb * yy.x, -0.5 * ihb * ihb * yy.y);   // (84)
        const cplx q_rqq_q = c_sub(c_make(dqCdq, 0), f6[1]);
        const cplx Q_rQQ_Q = c_sub(c_make(dQGdQ, 0), f6[2]);
        const cplx q_rqQ_Q = f6[3];
        const cplx Pq_dq = c_sub(c_make(piq_dq, 0), f6[4]);
        const cplx PQ_dQ = c_add(c_make(p0_dQ, 0), f6[5]);
        cplx ex = gamma;
        ex = c_sub(ex, c_scale(q_rqq_q, 0.5));
        ex = c_sub(ex, c_scale(Q_rQQ_Q, 0.5));
        ex = c_add(ex, q_rqQ_Q);
        ex = c_add(ex, c_mul(c_make(0.0, -ihb), Pq_dq));
        ex = c_add(ex, c_mul(c_make(0.0, ihb), PQ_dQ));
        cplx nacQ = c_make(0, 0), nacq = c_make(0, 0), nacqQ = c_make(0, 0);
        if (A.has_nac) {
            cplx f7[7] = {c_mul(hat[UN1], rho[WN1]), c_mul(hat[WN1], rho[Y]), c_mul(hat[UN1], rho[Y]), c_mul(hat[WDQ], rho[WN1]),
                          c_mul(hat[UDQ], rho[WN1]), c_mul(hat[UDQ], rho[UN1]), c_mul(hat[UN1], rho[WDQ])};
            sum_rows<DP>(f7, one);
            nacqQ = f7[0];
            const cplx PQ_n1 = c_add(c_make(W.p0n1, 0), f7[1]);
            const cplx Pq_n1 = c_sub(c_make(piq_n1, 0), f7[2]);
            nacQ = c_sub(c_make(dQGn1, 0), f7[3]);                               // dQ^T RQQ n1
            nacQ = c_sub(nacQ, f7[4]);                                           // - dq^T RqQ n1
            nacQ = c_add(nacQ, c_mul(c_make(0.0, -ihb), PQ_n1));
            nacQ.x += W.n2;
            nacq = c_sub(c_make(dqCn1, 0), f7[5]);                               // dq^T Rqq n1
            nacq = c_sub(nacq, f7[6]);                                           // - n1^T RqQ dQ
            nacq = c_add(nacq, c_mul(c_make(0.0, ihb), Pq_n1));
            nacq.x += W.n2;
        }
        const int weak_any = row_max(weak);       // the pivot lanes' verdicts, known to lane 0 of the trajectory

        // ---- hand-over to wm_tail_kernel (one THREAD per trajectory does the branch-tracked square roots, the exponentials
        //      and eqns (85), (100): in here that scalar code would run with 4 of 64 lanes active) ----
        if (r == 0 && active) {
            if (W.flags) {
                W.flags[tr] = weak_any;             // 1: left to wm_kernel<false> (full partial pivoting) in the same stream
                if (weak_any) atomicAdd(W.flags + n, 1);
            }
            double2 *out = (double2 *)(A.scratch + tr * WM_TAIL_FIELDS);
            out[0] = detA; out[1] = detM; out[2] = ex; out[3] = make_double2(eps, dqCdq);
            out[4] = make_double2(piq_dq, 0.0); out[5] = nacQ; out[6] = nacq; out[7] = nacqQ;
        }
    }
}

// One thread per trajectory: trackers of sqrt(detA), sqrt(detM) (propagators.py:1336, 1389), eqn (85) and (100), export
// of eqn (75); per-workgroup partial sums in a fixed order.  Trajectories flagged for the pivoted re-run are skipped.
__global__ __launch_bounds__(256) void wm_tail_kernel(WmArgs A) {
    __shared__ double red[4 * 4];
    const sc_wm_consts &W = A.wc;
    const double ihb = 1.0 / SC_HBAR;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t tr = (int64_t)blockIdx.x * 256 + threadIdx.x; tr < A.st.n; tr += (int64_t)gridDim.x * 256) {
        if (W.flags && W.flags[tr]) continue;
        const double2 *in = (const double2 *)(A.scratch + tr * WM_TAIL_FIELDS);
        const cplx detA = in[0], detM = in[1], ex = in[2], nacQ = in[5], nacq = in[6], nacqQ = in[7];
        const double eps = in[3].x, dqCdq = in[3].y, piq_dq = in[4].x;
        cplx *prevA = (cplx *)W.detA + tr, *prevM = (cplx *)W.detM + tr;
        const double sgA = wm_track_sign(A.track, detA, prevA, W.sgnA + tr);
        const double sgM = wm_track_sign(A.track, detM, prevM, W.sgnM + tr);
        const cplx cpre = c_scale(c_sqrt(((const cplx *)A.st.c2)[tr]), A.st.sgn[tr]);
        const cplx phase = c_exp(c_make(0.0, A.st.act[tr] * ihb));
        cplx pre = c_mul(cpre, phase);
        pre = c_mul(pre, c_scale(c_inv(c_sqrt(detA)), sgA));
        const double wgt = 1.0 / (A.mc_norm * A.probi[tr]);
        if (W.coef_out) {                   // eqn (75) without its x-dependent part, propagators.py:1408-1432
            const cplx v = c_mul(pre, c_exp(c_make(eps - 0.5 * dqCdq, -ihb * piq_dq)));
            ((cplx *)W.coef_out)[tr] = c_scale(v, W.pre_coef * wgt);
        }
        pre = c_mul(pre, c_scale(c_inv(c_sqrt(detM)), sgM));
        const cplx cq = c_scale(c_mul(pre, c_exp(ex)), W.pre * wgt);          // (85) / (n P (2 pi hbar)^D)
        acc[0] += cq.x; acc[1] += cq.y;
        if (A.cq_out) ((cplx *)A.cq_out)[tr] = cq;
        if (A.has_nac) {
            cplx kq = c_mul(c_add(nacqQ, c_mul(nacQ, nacq)), cq);             // (100)
            kq = c_scale(kq, ihb * ihb);
            acc[2] += kq.x; acc[3] += kq.y;
            if (A.kq_out) ((cplx *)A.kq_out)[tr] = kq;
        }
    }
    block_sum<4>(acc, red);
    if (threadIdx.x < 4) A.partials[(size_t)blockIdx.x * 4 + threadIdx.x] = acc[threadIdx.x];
}

template <int D, int DP>
int launch(const WmArgs &a, int grid, hipStream_t s) {
    const size_t lds = WmSmallLayout<D, DP>::bytes;
    if (hipFuncSetAttribute((const void *)wm_small_kernel<D, DP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
        return sc_check_launch("sc_wm_correlate (LDS attribute)");
    // persistent grid: two workgroups per CU are resident; every further one would stage the constants (16 KB) again
    const int resident = 2 * 256;
    hipLaunchKernelGGL((wm_small_kernel<D, DP>), dim3(grid < resident ? grid : resident), dim3(256), lds, s, a);
    int rc = sc_check_launch("sc_wm_correlate (register-resident kernel)");
    if (rc != SC_OK) return rc;
    hipLaunchKernelGGL(wm_tail_kernel, dim3(grid), dim3(256), 0, s, a);        // partials[0 .. grid)
    rc = sc_check_launch("sc_wm_correlate (scalar tails)");
    return rc == SC_OK ? 1 : rc;
}

}  // namespace

// instantiated shapes: full rank up to 8 modes, and the molecular cases of the reference's data (D = 3 N_atoms
// Cartesian coordinates, d' = D - 6 vibrations, D - 5 for linear molecules): methylium (12, 6)
int sc_wm_launch_small(const WmArgs &a, int grid, hipStream_t s) {
    const int D = a.st.dim, dp = a.wc.dprime;
#define SC_WM_CASE(D_, DP_) if (D == D_ && dp == DP_) return launch<D_, DP_>(a, grid, s);
    SC_WM_CASE(1, 1) SC_WM_CASE(2, 2) SC_WM_CASE(3, 3) SC_WM_CASE(4, 4) SC_WM_CASE(5, 5) SC_WM_CASE(6, 6)
    SC_WM_CASE(7, 7) SC_WM_CASE(8, 8) SC_WM_CASE(9, 3) SC_WM_CASE(12, 6)
    SC_WM_CASE(6, 1) SC_WM_CASE(9, 4) SC_WM_CASE(12, 7)        // diatomic, linear triatomic, linear four-atom molecule (d' = D - 5)
#undef SC_WM_CASE
    return 0;
}
