// Walton-Manolopoulos prefactor and correlation terms for SMALL matrices (D <= 16, e = 2 d' <= 16): register-resident.
//
// The same arithmetic as wm_kernel in sc_wm.hip (reference semiclassical/propagators.py:1132-1389 _expand_L /
// _prefactor, :1577-1614 eqn (85), :1652-1719 eqn (100)) with a different mapping to the machine:
//
//   * one trajectory per 16-lane DPP row, four trajectories per wavefront, sixteen per workgroup; lane r of a row holds
//     ROW r of every matrix of its trajectory in registers (D, d' are template parameters: all loops are unrolled and
//     every register index is static);
//   * products C = A B (A, B both per trajectory) take row k of B from lane k with a 64-bit DPP row_newbcast (the lane
//     is an immediate because k is a compile-time loop index); products with a constant matrix take the constant from
//     scalar registers (constant address space => s_load) and cost no cross-lane traffic at all;
//   * the two linear systems are solved by Gauss-Jordan elimination with TRUE partial pivoting where the pivot is a
//     LANE, not a register: the pivot row is picked by an integer-key DPP maximum over the unused lanes and broadcast
//     with ds_bpermute (the only data-dependent lane index in the kernel), so no register is ever indexed dynamically.
//     Nothing is inverted explicitly: A'^T Wm^T = BQ'^T is solved for the D right-hand sides the later formulas need
//     (57, 59), and M' rho = hat for the five vectors of the bilinear forms;
//   * LDS holds only the per-lane rows of the constants (staged once per workgroup) and one small exchange buffer per
//     trajectory for the two transpositions (columns of Mq', Mp'; rows of Wm);  there is no __syncthreads in the loop.
//
// Restatement (see sc_wm.hip for the derivation): e = 2d', Mq' = [Mqq U, Mqp U], Mp' = [Mpq U, Mpp U],
//   A' = Cst' + Mq'^T Gt Mq' + i/hbar (2G - H),  G = Mp'^T Mq',  H[i][j] = i < d' ? G[i][j] : G[j][i],
//   BQ' = Gt Mq' + i/hbar Mp',  Wm = BQ' A'^-1,  Gt~ = Gt - Wm BQ'^T (57),  Gti = Wm Bq'^T (59),
//   V = Gti iGi0,  CQQ = Gt~ - V Gti^T (70),  M' = U^T (G0 + CQQ) U,  rho_v = M'^-1 U^T v.
#include "sc_wm.h"
#include "sc_row16.h"

namespace {

// bytes of LDS: constants (doubles) + per-group exchange buffers (complex)
template <int D, int DP>
struct WmSmallLayout {
    static constexpr int E = 2 * DP, EP = E + 1;
    static constexpr int n_const = 5 * 16 * D        // rows of Gt, G0, Cqq, S, iGi0, zero padded to 16 lanes
                                   + 16 * D          // UT[i][a] = U[a][i], zero for i >= d'
                                   + 2 * 16 * E      // CstT[i][j] = Cst[j][i] (complex), zero for i >= e
                                   + 8 * 16;         // q0, p0, n1, s_n1, w_n1, crow = Cqq n1, 2 spare
    static constexpr int H = (D + 1) / 2;            // rows per half of the Wm exchange
    static constexpr int xh = H * EP;                // complex: transposition buffer (D x EP reals fit as well)
    static constexpr int xq = (E > D ? E : D) * D;   // complex: BQ'^T (e x D), later the rows of Gti (D x D)
    static constexpr int xbuf = xh + xq;             // complex values per trajectory
    static constexpr size_t bytes = (size_t)n_const * 8 + (size_t)16 * xbuf * 16 + 16 * 4 * 8;
};

#ifndef SC_WM_SMALL_OCC
// waves per SIMD the kernel is compiled for (register budget 512 / OCC).  Measured on MI355X, methylium (12, 6),
// n = 1e5: 1 -> 1.37 ms (the allocator parks idle rows in AGPRs), 2 -> 2.3 .. 2.9 ms (the same rows go to scratch).
#define SC_WM_SMALL_OCC 1
#endif

template <int D, int DP>
__global__ __launch_bounds__(256, SC_WM_SMALL_OCC) void wm_small_kernel(WmArgs A) {
    typedef WmSmallLayout<D, DP> L;
    constexpr int E = L::E, EP = L::EP, DD = D * D;
    extern __shared__ double2 smem2[];
    const int tid = threadIdx.x, r = tid & 15, grp = tid >> 4, rowbase = tid & 48;
    const sc_wm_consts &W = A.wc;
    const double ihb = 1.0 / SC_HBAR;

    double *ls = (double *)smem2;
    double *sGt = ls;   ls += 16 * D;
    double *sG0 = ls;   ls += 16 * D;
    double *sCqq = ls;  ls += 16 * D;
    double *sS = ls;    ls += 16 * D;
    double *siG = ls;   ls += 16 * D;
    double *sUT = ls;   ls += 16 * D;
    double *sCstT = ls; ls += 2 * 16 * E;
    double *cvec = ls;  ls += 8 * 16;            // [q0 | p0 | n1 | s_n1 | w_n1 | crow | - | -][16]
    cplx *xall = (cplx *)ls;
    double *red = (double *)(xall + 16 * L::xbuf);

    // ---- stage the per-lane rows of the constants (once per workgroup) ----
    for (int e = tid; e < 16 * D; e += 256) {
        const int i = e / D, b = e - i * D;
        const bool in = i < D;
        sGt[e] = in ? W.Gt[i * D + b] : 0.0;
        sG0[e] = in ? W.G0[i * D + b] : 0.0;
        sCqq[e] = in ? W.Cqq[i * D + b] : 0.0;
        sS[e] = in ? W.S[i * D + b] : 0.0;
        siG[e] = in ? W.iGi0[i * D + b] : 0.0;
        sUT[e] = i < DP ? W.U[b * DP + i] : 0.0;
    }
    for (int e = tid; e < 16 * E; e += 256) {
        const int i = e / E, j = e - i * E;
        sCstT[2 * e] = i < E ? W.Cst[2 * (j * E + i)] : 0.0;
        sCstT[2 * e + 1] = i < E ? W.Cst[2 * (j * E + i) + 1] : 0.0;
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

    kptr kU = (kptr)W.U, kGt = (kptr)W.Gt, kiG = (kptr)W.iGi0, kBq = (kptr)W.Bq;
    const double q0r = cvec[r], p0r = cvec[16 + r], n1r = cvec[32 + r], wn1r = cvec[64 + r], crowr = cvec[80 + r];
    cplx *xc = xall + grp * L::xbuf;        // exchange buffers of this trajectory: transpositions ...
    double *xr = (double *)xc;
    cplx *xq = xc + L::xh;                  // ... and group-uniform operands (BQ'^T, then the rows of Gti)

    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const int64_t n = A.st.n, stride = (int64_t)gridDim.x * 16;
    for (int64_t t0 = (int64_t)blockIdx.x * 16; t0 < n; t0 += stride) {
        const bool active = t0 + grp < n;
        const int64_t tr = active ? t0 + grp : n - 1;      // idle rows redo the last trajectory and discard it
        const double *qp = A.st.qp + tr * 2 * D, *zi = A.zi + tr * 2 * D;
        const double *M = A.st.mono + tr * 4 * (int64_t)DD;
        // the scalar loads of the constants are redone per trajectory next to their use: hoisted out of the loop they
        // would occupy (and spill) several hundred scalar registers
        asm volatile("" : "+s"(kU), "+s"(kGt), "+s"(kiG), "+s"(kBq));
        // the same for the per-lane rows of the constants in LDS: read where they are used, never kept across phases
        int lofs = 0;
        asm volatile("" : "+v"(lofs));
        const double *cGt = sGt + lofs, *cG0 = sG0 + lofs, *cCqq = sCqq + lofs, *cS = sS + lofs, *ciG = siG + lofs;
        const double *cUT = sUT + lofs, *cCstT = sCstT + lofs;

        // ---- rows of the monodromy blocks; Mq' = [Mqq U, Mqp U], Mp' = [Mpq U, Mpp U] (row r) ----
        double Mq[E], Mp[E];
        double qv = 0.0, pv = 0.0, dq = 0.0, dpv = 0.0;
        WM_BLOCK {
            double mqq[D], mqp[D], mpq[D], mpp[D];
#pragma unroll
            for (int b = 0; b < D; ++b) { mqq[b] = 0.0; mqp[b] = 0.0; mpq[b] = 0.0; mpp[b] = 0.0; }
            if (r < D) {
#pragma unroll
                for (int b = 0; b < D; ++b) {
                    mqq[b] = M[r * D + b]; mqp[b] = M[DD + r * D + b];
                    mpq[b] = M[2 * DD + r * D + b]; mpp[b] = M[3 * DD + r * D + b];
                }
                qv = qp[r]; pv = qp[D + r];
                dq = q0r - zi[r]; dpv = p0r - zi[D + r];
            }
#pragma unroll
            for (int j = 0; j < E; ++j) { Mq[j] = 0.0; Mp[j] = 0.0; }
            // one ROW of U (contiguous scalar loads) per block: the scalar registers hold d' constants at a time
            sfor_bb<0, D>([&](auto bcn) {
                constexpr int b = decltype(bcn)::value;
#pragma unroll
                for (int j = 0; j < DP; ++j) {
                    const double u = kU[b * DP + j];
                    Mq[j] = fma(mqq[b], u, Mq[j]); Mq[DP + j] = fma(mqp[b], u, Mq[DP + j]);
                    Mp[j] = fma(mpq[b], u, Mp[j]); Mp[DP + j] = fma(mpp[b], u, Mp[DP + j]);
                }
            });
        }
        const double dQ = r < D ? q0r - qv : 0.0;

        // ---- columns of Mq', Mp' (lane i < e holds column i) through the exchange buffer; TqT = (Gt Mq')^T ----
        double MqT[D], MpT[D], TqT[D];
        WM_BLOCK {
            if (r < D) {
#pragma unroll
                for (int j = 0; j < E; ++j) xr[r * EP + j] = Mq[j];
            }
            wave_lds_fence();
#pragma unroll
            for (int a = 0; a < D; ++a) MqT[a] = r < E ? xr[a * EP + r] : 0.0;
            wave_lds_fence();
            if (r < D) {
#pragma unroll
                for (int j = 0; j < E; ++j) xr[r * EP + j] = Mp[j];
            }
            wave_lds_fence();
#pragma unroll
            for (int a = 0; a < D; ++a) MpT[a] = r < E ? xr[a * EP + r] : 0.0;
            wave_lds_fence();
        }
        sfor_bb<0, D>([&](auto ac) {      // in-lane with the constant (symmetric) Gt, one row of it per block
            constexpr int a = decltype(ac)::value;
            double s = 0.0;
#pragma unroll
            for (int b = 0; b < D; ++b) s = fma(kGt[a * D + b], MqT[b], s);
            TqT[a] = s;
        });

        // ---- GT[i][j] = G[j][i] first (the last use of Mp' rows and Mq' columns), then G = Mp'^T Mq' (row i) and
        //      ST[i][j] = (Mq'^T Gt Mq')[j][i] ----
        double G[E], ST[E], GT[E];
#pragma unroll
        for (int j = 0; j < E; ++j) GT[j] = 0.0;
        sfor_bb<0, D>([&](auto ac) {
            constexpr int a = decltype(ac)::value;
#pragma unroll
            for (int j = 0; j < E; ++j) GT[j] = fma(MqT[a], bc<a>(Mp[j]), GT[j]);
        });
#pragma unroll
        for (int j = 0; j < E; ++j) { G[j] = 0.0; ST[j] = 0.0; }
        sfor_bb<0, D>([&](auto ac) {
            constexpr int a = decltype(ac)::value;
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const double x = bc<a>(Mq[j]);
                G[j] = fma(MpT[a], x, G[j]);
                ST[j] = fma(TqT[a], x, ST[j]);
            }
        });

        // ---- row i of (A'/s)^T and of the right-hand sides BQ'^T ----
        cplx At[E], Rh[D];
        WM_BLOCK {
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const double cre = cCstT[2 * (r * E + j)], cim = cCstT[2 * (r * E + j) + 1];
                const double im = j < DP ? GT[j] : 2.0 * GT[j] - G[j];
                At[j] = c_make((cre + ST[j]) * W.inv_scale_a, (cim + ihb * im) * W.inv_scale_a);
            }
#pragma unroll
            for (int a = 0; a < D; ++a) Rh[a] = c_make(TqT[a], ihb * MpT[a]);
            // BQ'^T also goes to LDS: eqn (57) reads it back as group-uniform operands after the elimination
            if (r < E) {
#pragma unroll
                for (int a = 0; a < D; ++a) xq[r * D + a] = Rh[a];
            }
        }

        int myk, src;
        cplx detA;
        gauss_jordan_rows<E, D>(At, Rh, r >= E, r, rowbase, myk, src, detA);

        // ---- Wm = BQ' A'^-1: the pivot lane of step k holds Wm[:, k] (scaled by s); write it as column k, read row r ----
        cplx Wm[E];
        sfor_bb<0, 2>([&](auto hc) {               // two halves of the rows: the exchange buffer holds (D+1)/2 of them
            constexpr int h0 = decltype(hc)::value ? L::H : 0, h1 = decltype(hc)::value ? D : L::H;
            if (r < E) {
#pragma unroll
                for (int a = h0; a < h1; ++a) xc[(a - h0) * EP + myk] = c_scale(Rh[a], W.inv_scale_a);
            }
            wave_lds_fence();
            if (r >= h0 && r < h1) {
#pragma unroll
                for (int k = 0; k < E; ++k) Wm[k] = xc[(r - h0) * EP + k];
            }
            wave_lds_fence();
        });
        if (r >= D) {
#pragma unroll
            for (int k = 0; k < E; ++k) Wm[k] = c_make(0.0, 0.0);
        }

        // ---- Gt~ = Gt - Wm BQ'^T (57) ----
        cplx Gtl[D], Gti[D];
#pragma unroll
        for (int b = 0; b < D; ++b) Gtl[b] = c_make(cGt[r * D + b], 0.0);
        sfor_bb<0, E>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
#pragma unroll
            for (int b = 0; b < D; ++b) Gtl[b] = c_fnma(Wm[k], xq[k * D + b], Gtl[b]);
        });
        // ---- Gti = Wm Bq'^T (59);  Bq' = [Gamma_i U | -i/hbar U]: real in its first d' columns, imaginary in the last d'
        sfor_bb<0, D>([&](auto bcn) {
            constexpr int b = decltype(bcn)::value;
            cplx s = c_make(0.0, 0.0);
#pragma unroll
            for (int k = 0; k < DP; ++k) {
                const double br = kBq[2 * (b * E + k)], bi = kBq[2 * (b * E + DP + k) + 1];
                s.x = fma(Wm[k].x, br, s.x); s.y = fma(Wm[k].y, br, s.y);
                s.x = fma(-Wm[DP + k].y, bi, s.x); s.y = fma(Wm[DP + k].x, bi, s.y);
            }
            Gti[b] = s;
        });

        // ---- per-trajectory vectors: g = iGi0 (p0 - p_i), s_dq = S dq, w_dQ = G0 dQ, cdq = Cqq dq, g0g = G0 g ----
        double gv = 0.0, sdq = 0.0, wdQ = 0.0, cdq = 0.0, g0g = 0.0;
        cplx y = c_make(r < D ? pv - p0r : 0.0, 0.0), u1 = c_make(0.0, 0.0), u2 = c_make(0.0, 0.0);
        WM_BLOCK {
            sfor<0, D>([&](auto bcn) {
                constexpr int b = decltype(bcn)::value;
                const double xdp = bc<b>(dpv), xdq = bc<b>(dq), xdQ = bc<b>(dQ);
                gv = fma(ciG[r * D + b], xdp, gv);
                sdq = fma(cS[r * D + b], xdq, sdq);
                cdq = fma(cCqq[r * D + b], xdq, cdq);
                wdQ = fma(cG0[r * D + b], xdQ, wdQ);
            });
        }
        WM_BLOCK {
            sfor<0, D>([&](auto bcn) {
                constexpr int b = decltype(bcn)::value;
                const double xg = bc<b>(gv), xs = bc<b>(sdq), xn = cvec[48 + b];
                g0g = fma(cG0[r * D + b], xg, g0g);
                y.x = fma(Gti[b].x, xg, y.x); y.y = fma(Gti[b].y, xg, y.y);
                u1.x = fma(Gti[b].x, xs, u1.x); u1.y = fma(Gti[b].y, xs, u1.y);
                u2.x = fma(Gti[b].x, xn, u2.x); u2.y = fma(Gti[b].y, xn, u2.y);
            });
        }

        // ---- V = Gti iGi0 ; CQQ = Gt~ - V Gti^T (70), in place in Gtl.  The rows of Gti go to the exchange buffer and
        //      come back as group-uniform (broadcast) LDS reads: their registers are free while the product runs ----
        {
            cplx V[D];
            sfor_bb<0, D>([&](auto bcn) {          // iGi0 is symmetric: column b = row b, contiguous scalar loads
                constexpr int b = decltype(bcn)::value;
                cplx s = c_make(0.0, 0.0);
#pragma unroll
                for (int k = 0; k < D; ++k) { const double x = kiG[b * D + k]; s.x = fma(Gti[k].x, x, s.x); s.y = fma(Gti[k].y, x, s.y); }
                V[b] = s;
            });
            WM_BLOCK {
                wave_lds_fence();
                if (r < D) {
#pragma unroll
                    for (int k = 0; k < D; ++k) xq[r * D + k] = Gti[k];
                }
                wave_lds_fence();
            }
            sfor_bb<0, D>([&](auto bcn) {
                constexpr int b = decltype(bcn)::value;
                cplx s = c_make(0.0, 0.0);
#pragma unroll
                for (int k = 0; k < D; ++k) s = c_fma(V[k], xq[b * D + k], s);
                Gtl[b] = c_sub(Gtl[b], s);
            });
            wave_lds_fence();
        }
        if (W.cqq_out && active && r < D) {
            cplx *out = (cplx *)W.cqq_out + tr * (int64_t)DD + r * D;
#pragma unroll
            for (int b = 0; b < D; ++b) out[b] = Gtl[b];
        }
        if (W.dvec_out && active && r < D)       // C_qQ^T (q0 - q) + i/hbar PI_Q = u_dq + i/hbar (y + p0)
            ((cplx *)W.dvec_out)[tr * (int64_t)D + r] = c_make(u1.x - ihb * y.y, u1.y + ihb * (y.x + p0r));

        // ---- M'/(2 pi) = U^T (G0 + CQQ) U / (2 pi) (row i < d') and hat_v = U^T {u_dq, u_n1, w_dQ, w_n1, y} ----
        cplx Mr[DP], hat[5], R[DP];
#pragma unroll
        for (int j = 0; j < DP; ++j) R[j] = c_make(0.0, 0.0);
        sfor_bb<0, D>([&](auto bcn) {
            constexpr int b = decltype(bcn)::value;
            const double gre = cG0[r * D + b] + Gtl[b].x;
#pragma unroll
            for (int j = 0; j < DP; ++j) {
                const double u = kU[b * DP + j];
                R[j].x = fma(gre, u, R[j].x); R[j].y = fma(Gtl[b].y, u, R[j].y);
            }
        });
#pragma unroll
        for (int j = 0; j < DP; ++j) Mr[j] = c_make(0.0, 0.0);
#pragma unroll
        for (int v = 0; v < 5; ++v) hat[v] = c_make(0.0, 0.0);
        sfor_bb<0, D>([&](auto ac) {
            constexpr int a = decltype(ac)::value;
            const double ui = cUT[r * D + a];
#pragma unroll
            for (int j = 0; j < DP; ++j) {
                Mr[j].x = fma(ui, bc<a>(R[j].x), Mr[j].x); Mr[j].y = fma(ui, bc<a>(R[j].y), Mr[j].y);
            }
            hat[0].x = fma(ui, bc<a>(u1.x), hat[0].x); hat[0].y = fma(ui, bc<a>(u1.y), hat[0].y);
            hat[1].x = fma(ui, bc<a>(u2.x), hat[1].x); hat[1].y = fma(ui, bc<a>(u2.y), hat[1].y);
            hat[2].x = fma(ui, bc<a>(wdQ), hat[2].x);
            hat[3].x = fma(ui, bc<a>(wn1r), hat[3].x);
            hat[4].x = fma(ui, bc<a>(y.x), hat[4].x); hat[4].y = fma(ui, bc<a>(y.y), hat[4].y);
        });
#pragma unroll
        for (int j = 0; j < DP; ++j) Mr[j] = c_scale(Mr[j], W.inv_two_pi);
        cplx sol[5];
#pragma unroll
        for (int v = 0; v < 5; ++v) sol[v] = hat[v];
        cplx detM;
        gauss_jordan_rows<DP, 5>(Mr, sol, r >= DP, r, rowbase, myk, src, detM);
        // rho_v[k] sits in the pivot lane of step k: fetch it into lane k;  rho = M'^-1 hat (the 1/2pi of the scaling)
        cplx rho[5];
#pragma unroll
        for (int v = 0; v < 5; ++v) rho[v] = c_scale(perm(src, sol[v]), W.inv_two_pi);

        // ---- bilinear forms a^T iM b and the scalar sums over the modes ----
        enum { UDQ = 0, UN1 = 1, WDQ = 2, WN1 = 3, Y = 4 };
        auto form = [&](int a, int b) {
            const cplx t = r < DP ? c_mul(hat[a], rho[b]) : c_make(0.0, 0.0);
            return c_make(row_sum(t.x), row_sum(t.y));
        };
        const double piq = p0r - g0g;                                            // (72)
        const double dqCdq = row_sum(dq * cdq), dqCn1 = row_sum(dq * crowr), dQGdQ = row_sum(dQ * wdQ);
        const double dQGn1 = row_sum(dQ * wn1r), piq_dq = row_sum(piq * dq), piq_n1 = row_sum(piq * n1r);
        const double p0_dQ = row_sum(p0r * dQ);
        const double eps = -0.5 * ihb * ihb * row_sum(dpv * gv);                 // (74), b0 = 0
        const cplx yy = form(Y, Y);
        const cplx gamma = c_make(eps - 0.5 * ihb * ihb * yy.x, -0.5 * ihb * ihb * yy.y);   // (84)
        const cplx q_rqq_q = c_sub(c_make(dqCdq, 0), form(UDQ, UDQ));
        const cplx Q_rQQ_Q = c_sub(c_make(dQGdQ, 0), form(WDQ, WDQ));
        const cplx q_rqQ_Q = form(UDQ, WDQ);
        const cplx Pq_dq = c_sub(c_make(piq_dq, 0), form(UDQ, Y));
        const cplx PQ_dQ = c_add(c_make(p0_dQ, 0), form(WDQ, Y));
        cplx ex = gamma;
        ex = c_sub(ex, c_scale(q_rqq_q, 0.5));
        ex = c_sub(ex, c_scale(Q_rQQ_Q, 0.5));
        ex = c_add(ex, q_rqQ_Q);
        ex = c_add(ex, c_mul(c_make(0.0, -ihb), Pq_dq));
        ex = c_add(ex, c_mul(c_make(0.0, ihb), PQ_dQ));
        cplx nacQ = c_make(0, 0), nacq = c_make(0, 0), nacqQ = c_make(0, 0);
        if (A.has_nac) {
            nacqQ = form(UN1, WN1);
            const cplx PQ_n1 = c_add(c_make(W.p0n1, 0), form(WN1, Y));
            const cplx Pq_n1 = c_sub(c_make(piq_n1, 0), form(UN1, Y));
            nacQ = c_sub(c_make(dQGn1, 0), form(WDQ, WN1));                      // dQ^T RQQ n1
            nacQ = c_sub(nacQ, form(UDQ, WN1));                                  // - dq^T RqQ n1
            nacQ = c_add(nacQ, c_mul(c_make(0.0, -ihb), PQ_n1));
            nacQ.x += W.n2;
            nacq = c_sub(c_make(dqCn1, 0), form(UDQ, UN1));                      // dq^T Rqq n1
            nacq = c_sub(nacq, form(UN1, WDQ));                                  // - n1^T RqQ dQ
            nacq = c_add(nacq, c_mul(c_make(0.0, ihb), Pq_n1));
            nacq.x += W.n2;
        }

        // ---- one lane per trajectory: branch-tracked square roots, eqns (85) and (100) ----
        if (r == 0 && active) {
            cplx *prevA = (cplx *)W.detA + tr, *prevM = (cplx *)W.detM + tr;
            const double sA = wm_track_sign(A.track, detA, prevA, W.sgnA + tr);
            const double sM = wm_track_sign(A.track, detM, prevM, W.sgnM + tr);
            const cplx cpre = c_scale(c_sqrt(((const cplx *)A.st.c2)[tr]), A.st.sgn[tr]);
            const cplx phase = c_exp(c_make(0.0, A.st.act[tr] * ihb));
            cplx pre = c_mul(cpre, phase);
            pre = c_mul(pre, c_scale(c_inv(c_sqrt(detA)), sA));
            const double wgt = 1.0 / (A.mc_norm * A.probi[tr]);
            if (W.coef_out) {                   // eqn (75) without its x-dependent part, propagators.py:1408-1432
                const cplx v = c_mul(pre, c_exp(c_make(eps - 0.5 * dqCdq, -ihb * piq_dq)));
                ((cplx *)W.coef_out)[tr] = c_scale(v, W.pre_coef * wgt);
            }
            pre = c_mul(pre, c_scale(c_inv(c_sqrt(detM)), sM));
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
    }
    // ---- per-workgroup partial sums, fixed order ----
    if (r == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red[grp * 4 + i] = acc[i];
    }
    __syncthreads();
    if (tid < 4) {
        double s = 0.0;
        for (int g = 0; g < 16; ++g) s += red[g * 4 + tid];
        A.partials[(size_t)blockIdx.x * 4 + tid] = s;
    }
}

template <int D, int DP>
int launch(const WmArgs &a, int grid, hipStream_t s) {
    const size_t lds = WmSmallLayout<D, DP>::bytes;
    if (hipFuncSetAttribute((const void *)wm_small_kernel<D, DP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
        return sc_check_launch("sc_wm_correlate (LDS attribute)");
    hipLaunchKernelGGL((wm_small_kernel<D, DP>), dim3(grid), dim3(256), lds, s, a);
    const int rc = sc_check_launch("sc_wm_correlate (register-resident kernel)");
    return rc == SC_OK ? 1 : rc;
}

}  // namespace

// instantiated shapes: full rank up to 8 modes, and the molecular cases of the reference's data (D = 3 N_atoms
// Cartesian coordinates, d' = D - 6 vibrations): methylium (12, 6)
int sc_wm_launch_small(const WmArgs &a, int grid, hipStream_t s) {
    const int D = a.st.dim, dp = a.wc.dprime;
#define SC_WM_CASE(D_, DP_) if (D == D_ && dp == DP_) return launch<D_, DP_>(a, grid, s);
    SC_WM_CASE(1, 1) SC_WM_CASE(2, 2) SC_WM_CASE(3, 3) SC_WM_CASE(4, 4) SC_WM_CASE(5, 5) SC_WM_CASE(6, 6)
    SC_WM_CASE(7, 7) SC_WM_CASE(8, 8) SC_WM_CASE(9, 3) SC_WM_CASE(12, 6)
#undef SC_WM_CASE
    return 0;
}
