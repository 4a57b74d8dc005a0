// The caller loop of the reference (cli.py:401-436: autocorrelation, ic_correlation, step -- nt times) as ONE launch for a
// CONSTANT dense Hessian (MolecularHarmonicPotential, potentials.py:553-593) and small matrices (D <= 16): the methylium
// example of the reference (12 Cartesian coordinates, rank-6 widths) with the Herman-Kluk propagator.
//
// Step by step this loop is hk_correlate_rows16_kernel + reduce + hk_step_lin_kernel + guard per time step, and the step
// kernel is bound by reading and writing the 4 D^2 monodromy doubles of every trajectory (9.6 KB at D = 12) each step.
// Here a trajectory is loaded ONCE into the registers of its 16-lane row -- lane a holds row a of the four monodromy blocks
// (96 registers at D = 12), q_a, p_a -- runs all nsteps steps there and is written back once; per step only five partial
// sums per wavefront leave the chip (the scheme of sc_hk_run_sep16.hip).
//
// Per step and trajectory, in the order of the caller loop, with the arithmetic (and operation order) of the step-at-a-time
// kernels, so that both paths agree to rounding:
//   terms of C_auto, k_ic from the CURRENT state    hk_correlate_rows16_kernel: y = A dq, B dp, C dp as fused broadcast
//                                                   multiply-adds, sums over the modes, scalar tail   (propagators.py:784-911)
//   RK4 stages of (q, p), action, <T+V> at k4       hk_step_lin_kernel                                 (:86-119, 313-383)
//   [X; Y] <- Phi(dt) [X; Y]                        one product with the step matrix (sc_potential.lin_prop), half by half
//   prefactor matrix, determinant, branch tracker   real sandwiches L (M R) (sc_hk_consts.real_lr), fixed pivot order; a weak
//                                                   pivot repeats the elimination with the pivot searched among the lanes
//                                                   (the rows are still in registers)                  (:969-1052)
#include "sc_hk_run.h"
#include "sc_row16.h"

namespace {

template <int D, int DP, bool DIAG>
struct RunLinLayout {
    // per-lane rows of the constants, zero padded to 16 lanes; odd pitches spread the lanes over the LDS banks
    static constexpr int PH = D + 1, PP = 4 * D + 1, PL = D + 1, PR = DP + 1, PD = D + 1;
    static constexpr int n_doubles = 16 * PH + 16 * PP + 12 * 16 + (DIAG ? 0 : 2 * 16 * PL + 2 * 16 * PR) + 3 * 16 * PD;
    static constexpr size_t bytes = (size_t)n_doubles * 8;
};

#ifndef SC_RUNLIN_OCC
#define SC_RUNLIN_OCC 2
#endif
#ifndef SC_RUNLIN_MODAL_OCC
#define SC_RUNLIN_MODAL_OCC 2
#endif
// MODAL (round 4, sc_hk_run_modal): the monodromy blocks of the state are in NORMAL-MODE coordinates (the caller transformed them and
// the prefactor constants), where the step matrix is 2 x 2 per mode: row a of the blocks is multiplied by (phi_qq, phi_qp; phi_pq,
// phi_pp)_a -- 8 D plain multiply-adds per lane and step instead of the 8 D^2 broadcast multiply-adds of the product with Phi.
template <int D, int DP, bool DIAG, bool MODAL>
__global__ __launch_bounds__(256, MODAL ? SC_RUNLIN_MODAL_OCC : SC_RUNLIN_OCC) void hk_run_lin_kernel(RunArgs R) {
    typedef RunLinLayout<D, DP, DIAG> L;
    constexpr int W = 2 * D, DD = D * D, N = DIAG ? D : DP;
    constexpr int PH = L::PH, PP = L::PP, PL = L::PL, PR = L::PR, PD = L::PD;
    extern __shared__ double2 smem2[];
    const StepArgs &A = R.step;
    const int tid = threadIdx.x, lane = tid & 63, r = tid & 15, grp = tid >> 4, wave = tid >> 6, rowbase = tid & 48;
    const double dt = A.dt, hh = 0.5 * dt, h6 = dt / 6.0;
    const bool nac = R.has_nac != 0;

    double *ls = (double *)smem2;
    double *sH = ls;    ls += 16 * PH;
    double *sPhi = ls;  ls += 16 * PP;             // row a: Phi_qq[a][:], Phi_qp[a][:], Phi_pq[a][:], Phi_pp[a][:]
    double *svec = ls;  ls += 12 * 16;             // x0, g0, 1/m, st, 1/st, qk, pk, nq0, np0, nrn, ngn, -
    double *sL1 = ls, *sL2 = sL1 + (DIAG ? 0 : 16 * PL), *sR1 = sL2 + (DIAG ? 0 : 16 * PL), *sR2 = sR1 + (DIAG ? 0 : 16 * PR);
    ls = sR2 + (DIAG ? 0 : 16 * PR);
    double *sOA = ls, *sOB = sOA + 16 * PD, *sOC = sOB + 16 * PD;

    for (int e = tid; e < 16 * D; e += 256) {
        const int i = e / D, b = e - i * D;
        const bool in = i < D;
        sH[i * PH + b] = in ? A.pot.par2[i * D + b] : 0.0;
        if (!DIAG) {
            sL1[i * PL + b] = i < DP ? A.hk.L1[2 * (i * D + b)] : 0.0;          // real parts (sc_hk_consts.real_lr)
            sL2[i * PL + b] = i < DP ? A.hk.L2[2 * (i * D + b)] : 0.0;
        }
        // overlap matrices: dense rows, or the diagonal spread into rows of zeros
        if (R.oc.diag) {
            sOA[i * PD + b] = (in && i == b) ? R.oc.A[i] : 0.0;
            sOB[i * PD + b] = (in && i == b) ? R.oc.B[i] : 0.0;
            sOC[i * PD + b] = (in && i == b) ? R.oc.C[i] : 0.0;
        } else {
            sOA[i * PD + b] = in ? R.oc.A[i * D + b] : 0.0;
            sOB[i * PD + b] = in ? R.oc.B[i * D + b] : 0.0;
            sOC[i * PD + b] = in ? R.oc.C[i * D + b] : 0.0;
        }
    }
    if (!DIAG) {
        for (int e = tid; e < 16 * DP; e += 256) {
            const int i = e / DP, j = e - i * DP;
            sR1[i * PR + j] = i < D ? A.hk.R1[2 * (i * DP + j)] : 0.0;
            sR2[i * PR + j] = i < D ? A.hk.R2[2 * (i * DP + j)] : 0.0;
        }
    }
    if (MODAL) {
        for (int e = tid; e < 16 * 4; e += 256) {
            const int i = e >> 2, k = e & 3;                                              // (phi_qq, phi_qp, phi_pq, phi_pp) of mode i
            sPhi[i * PP + k] = i < D ? R.mode_prop[i * 4 + k] : (k == 0 || k == 3 ? 1.0 : 0.0);
        }
    } else {
        for (int e = tid; e < 16 * 4 * D; e += 256) {
            const int i = e / (4 * D), k = e - i * 4 * D, blk = k / D, g = k - blk * D;      // blk: qq, qp, pq, pp
            const int row = (blk >> 1) * D + i, col = (blk & 1) * D + g;
            sPhi[i * PP + k] = i < D ? A.pot.lin_prop[row * W + col] : 0.0;
        }
    }
    if (tid < 16) {
        const bool in = tid < D;
        svec[tid] = in ? A.pot.par0[tid] : 0.0;
        svec[16 + tid] = in ? A.pot.par1[tid] : 0.0;
        svec[32 + tid] = in ? A.pot.inv_mass[tid] : 0.0;
        const double st = (DIAG && in) ? A.hk.st[tid] : 1.0;
        svec[48 + tid] = st; svec[64 + tid] = 1.0 / st;
        svec[80 + tid] = in ? R.oc.qk[tid] : 0.0;
        svec[96 + tid] = in ? R.oc.pk[tid] : 0.0;
        svec[112 + tid] = (in && nac) ? R.nc.q0[tid] : 0.0;
        svec[128 + tid] = (in && nac) ? R.nc.p0[tid] : 0.0;
        svec[144 + tid] = (in && nac) ? R.nc.rn[tid] : 0.0;
        svec[160 + tid] = (in && nac) ? R.nc.gn[tid] : 0.0;
    }
    __syncthreads();
    const double x0 = svec[r], g0 = svec[16 + r], im = svec[32 + r], sta = svec[48 + r], ista = svec[64 + r];
    const double qk = svec[80 + r], pk = svec[96 + r];
    kptr ksi = (kptr)A.hk.si;

    const int64_t n = A.st.n, stride = (int64_t)gridDim.x * 16;
    const int slot = blockIdx.x * 4 + wave;
    double one = 1.0;
    asm volatile("" : "+v"(one));

    for (int64_t t0 = (int64_t)blockIdx.x * 16; t0 < n; t0 += stride) {
        const bool active = t0 + grp < n, head = active && r == 0;
        const int64_t tr = active ? t0 + grp : n - 1;          // idle rows shadow the last trajectory and contribute nothing
        double *M = A.st.mono + tr * 4 * (int64_t)DD, *qp = A.st.qp + tr * 2 * D;
        // ---- the trajectory: rows of the four blocks as [half][q | p block][column]: half 0 = (Mqq, Mpq), half 1 = (Mqp, Mpp)
        double cur[2][2][D];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int b = 0; b < D; ++b) {
                cur[h][0][b] = r < D ? M[(h ? DD : 0) + r * D + b] : 0.0;
                cur[h][1][b] = r < D ? M[(h ? DD : 0) + 2 * DD + r * D + b] : 0.0;
            }
        double q = r < D ? qp[r] : 0.0, p = r < D ? qp[D + r] : 0.0;
        double S = A.st.act[tr], sgn = A.st.sgn[tr];
        cplx c2 = ((const cplx *)A.st.c2)[tr];
        const cplx vi = ((const cplx *)R.vi)[tr];
        const cplx nacq = nac ? ((const cplx *)R.nacq)[tr] : c_make(0.0, 0.0);
        const double wgt = 1.0 / (R.mc_norm * R.probi[tr]);

        for (int k = 0; k < R.nsteps; ++k) {
            int lofs = 0;
            asm volatile("" : "+v"(lofs));         // the per-lane constants are re-read from LDS every step, never kept across steps
            const double *cH = sH + lofs, *cPhi = sPhi + lofs, *cv = svec + lofs;
            double v5[5];
            // ---- terms of the correlation functions from the current state (hk_correlate_rows16_kernel) ----
            {
                const double *cA = sOA + lofs + r * PD, *cB = sOB + lofs + r * PD, *cC = sOC + lofs + r * PD;
                double dq = qk - q, dpp = pk - p;
                if (r >= D) { dq = 0.0; dpp = 0.0; }
                double ya = 0.0, yb = 0.0, yc = 0.0;
                dpp_guard(dq, dpp);
                // one matrix after the other (a row of constants is 2 D registers, and 4 D^2 / 16 * 2 are taken by the trajectory)
                auto times_row = [&](double &y, const double &x, const double *row) {
                    double rw[D];
#pragma unroll
                    for (int b = 0; b < D; ++b) rw[b] = row[b];
                    sfor<0, D>([&](auto bcn) { fmac_bc<decltype(bcn)::value>(y, x, rw[decltype(bcn)::value]); });
                };
                times_row(ya, dq, cA); times_row(yb, dpp, cB); times_row(yc, dpp, cC);
                const double nq0 = cv[112 + r], np0 = cv[128 + r], nrn = cv[144 + r], ngn = cv[160 + r];
                double t[6] = {dq * ya, dpp * yb, pk * dq, dq * yc, (nq0 - q) * nrn, (p - np0) * ngn}, s[6] = {0, 0, 0, 0, 0, 0};
                dpp_guard(t);
                sfor<0, D>([&](auto kc) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) fmac_bc<decltype(kc)::value>(s[i], t[i], one);
                });
                const cplx ex = c_make(-0.5 * s[0] - 0.5 / (SC_HBAR * SC_HBAR) * s[1], (-s[2] + s[3]) / SC_HBAR);
                const cplx vt = c_scale(c_exp(ex), R.oc.fac);
                const cplx c = c_scale(c_sqrt(c2), sgn);
                const cplx ph = c_exp(c_make(0.0, S / SC_HBAR));
                cplx cq = c_mul(c_mul(c_conj(vt), vi), c_mul(c, ph));
                cq = c_scale(cq, wgt);
                v5[0] = cq.x; v5[1] = cq.y; v5[2] = 0.0; v5[3] = 0.0;
                if (nac) {
                    const cplx nacQ = c_make(R.nc.n2 + s[4], -(R.nc.p0n1 + s[5]) / SC_HBAR);
                    cplx kq = c_mul(c_mul(nacQ, nacq), cq);
                    kq = c_scale(kq, 1.0 / (SC_HBAR * SC_HBAR));
                    v5[2] = kq.x; v5[3] = kq.y;
                }
            }
            // ---- (q, p, S): the explicit RK4 stages (V = E0 + g.dr + 1/2 dr.H.dr - origin, grad = g + H.dr), hk_step_lin_kernel ----
            {
                double qs = q, ps = p, kqs = 0.0, kps = 0.0, qn = 0.0, pn = 0.0, red5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
                double hrow[D];
#pragma unroll
                for (int b = 0; b < D; ++b) hrow[b] = cH[r * PH + b];
                sfor<0, 4>([&](auto sc_) {
                    constexpr int s = decltype(sc_)::value;
                    if (s > 0) { const double c = (s == 3) ? dt : hh; qs = q + c * kqs; ps = p + c * kps; }
                    double dr = r < D ? qs - x0 : 0.0;
                    double hd = 0.0, hd2 = 0.0;
                    dpp_guard(dr);
                    sfor<0, D>([&](auto bc_) {
                        constexpr int b = decltype(bc_)::value;
                        if (b & 1) fmac_bc<b>(hd2, dr, hrow[b]); else fmac_bc<b>(hd, dr, hrow[b]);
                    });
                    hd += hd2;
                    const double v = dr * g0 + 0.5 * dr * hd;
                    const double kq = ps * im, kp = -(g0 + hd), t = 0.5 * ps * ps * im;
                    red5[s] = t - v;
                    if (s == 3) red5[4] = t + v;
                    const double w = (s == 0 || s == 3) ? 1.0 : 2.0;
                    qn += w * kq; pn += w * kp;
                    kqs = kq; kps = kp;
                });
                double s5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
                dpp_guard(red5);
                sfor<0, D>([&](auto kc) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) fmac_bc<decltype(kc)::value>(s5[i], red5[i], one);
                });
#pragma unroll
                for (int s = 0; s < 4; ++s) s5[s] -= A.pot.scalar0;
                s5[4] += A.pot.scalar0;
                if (r < D) { q = q + h6 * qn; p = p + h6 * pn; }
                S = S + h6 * (s5[0] + 2.0 * s5[1] + 2.0 * s5[2] + s5[3]);
                v5[4] = s5[4];
            }
            // ---- [X; Y] <- Phi [X; Y], one half (Mqq, Mpq | Mqp, Mpp) at a time: row g of the old blocks comes from lane g
            //      inside the multiply-add, Phi[r][g] from LDS ----
            if constexpr (MODAL) {
                const double fqq = cPhi[r * PP], fqp = cPhi[r * PP + 1], fpq = cPhi[r * PP + 2], fpp = cPhi[r * PP + 3];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int b = 0; b < D; ++b) {
                        const double tq = cur[h][0][b], tp = cur[h][1][b];
                        cur[h][0][b] = fma(fqp, tp, fqq * tq);
                        cur[h][1][b] = fma(fpp, tp, fpq * tq);
                    }
            } else
            sfor<0, 2>([&](auto hc) {
                constexpr int h = decltype(hc)::value;
                double (&Tq)[D] = cur[h][0], (&Tp)[D] = cur[h][1];
                double Xq[D], Xp[D];
#pragma unroll
                for (int b = 0; b < D; ++b) { Xq[b] = 0.0; Xp[b] = 0.0; }
                dpp_guard(Tq, Tp);
                sfor<0, D>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    const double fqq = cPhi[r * PP + g], fqp = cPhi[r * PP + D + g];
                    const double fpq = cPhi[r * PP + 2 * D + g], fpp = cPhi[r * PP + 3 * D + g];
#pragma unroll
                    for (int b = 0; b < D; ++b) { fmac_bc<g>(Xq[b], Tq[b], fqq); fmac_bc<g>(Xp[b], Tq[b], fpq); }
#pragma unroll
                    for (int b = 0; b < D; ++b) { fmac_bc<g>(Xq[b], Tp[b], fqp); fmac_bc<g>(Xp[b], Tp[b], fpp); }
                });
#pragma unroll
                for (int b = 0; b < D; ++b) { Tq[b] = Xq[b]; Tp[b] = Xp[b]; }
            });
            // ---- prefactor matrix from the new rows (the formulas and the order of hk_step_lin_kernel) ----
            cplx mat[N];
            auto build_mat = [&]() {
                if constexpr (DIAG) {
#pragma unroll
                    for (int b = 0; b < D; ++b) {
                        const double sib = ksi[b], isib = 1.0 / sib;
                        mat[b] = r < D ? c_make(0.5 * (sta * isib * cur[0][0][b]), 0.5 * ((1.0 / SC_HBAR) * ista * isib * cur[0][1][b]))
                                       : c_make(0.0, 0.0);
                        mat[b] = r < D ? c_make(mat[b].x + 0.5 * (ista * sib * cur[1][1][b]), mat[b].y + 0.5 * (-SC_HBAR * sta * sib * cur[1][0][b]))
                                       : c_make(0.0, 0.0);
                    }
                } else {
                    double s1[DP], s2[DP], t1[DP], t2[DP];     // Mqq R1, Mqp R2, Mpp R2, Mpq R1
                    sfor<0, 2>([&](auto hc) {
                        constexpr int h = decltype(hc)::value;
                        double rr[DP];
                        double (&uq)[DP] = h ? s2 : s1, (&up)[DP] = h ? t1 : t2;
#pragma unroll
                        for (int j = 0; j < DP; ++j) { rr[j] = ((h ? sR2 : sR1) + lofs)[r * PR + j]; uq[j] = 0.0; up[j] = 0.0; }
                        dpp_guard(rr);
                        sfor<0, D>([&](auto bcn) {
                            constexpr int b = decltype(bcn)::value;
#pragma unroll
                            for (int j = 0; j < DP; ++j) { fmac_bc<b>(uq[j], rr[j], cur[h][0][b]); fmac_bc<b>(up[j], rr[j], cur[h][1][b]); }
                        });
                    });
                    cplx X1[DP], X2[DP];
#pragma unroll
                    for (int j = 0; j < DP; ++j) {
                        X1[j] = c_make(s1[j], -SC_HBAR * s2[j]);                 // Mqq R1 - i hbar Mqp R2
                        X2[j] = c_make(t1[j], (1.0 / SC_HBAR) * t2[j]);          // Mpp R2 + i/hbar Mpq R1
                        mat[j] = c_make(0.0, 0.0);
                    }
                    dpp_guard(X1, X2);
                    sfor<0, D>([&](auto ac) {
                        constexpr int a = decltype(ac)::value;
                        const double l1 = (sL1 + lofs)[r * PL + a], l2 = (sL2 + lofs)[r * PL + a];
#pragma unroll
                        for (int j = 0; j < DP; ++j) { fmac_bc<a>(mat[j].x, X1[j].x, l1); fmac_bc<a>(mat[j].y, X1[j].y, l1); }
#pragma unroll
                        for (int j = 0; j < DP; ++j) { fmac_bc<a>(mat[j].x, X2[j].x, l2); fmac_bc<a>(mat[j].y, X2[j].y, l2); }
                    });
#pragma unroll
                    for (int j = 0; j < DP; ++j) mat[j] = c_scale(mat[j], 0.5);
                }
            };
            build_mat();
            int weak = 0;
            cplx det = det_rows_fixed_order<N>(mat, r, weak);
            if (__builtin_amdgcn_readfirstlane(wave_max_i32(weak)) != 0) {
                // a trajectory of this wavefront met a weak pivot: its determinant by elimination with the pivot searched among
                // the lanes (what the fix-up launch of the step-at-a-time path does), from the rows that are still here
                build_mat();
                int myk, src;
                cplx det2, dummy[1] = {c_make(0.0, 0.0)};
                gauss_jordan_rows<N, 1>(mat, dummy, r >= N, r, rowbase, myk, src, det2);
                if (weak) det = det2;
            }
            if (c2.x < 0.0 && det.x < 0.0 && c2.y * det.y < 0.0) sgn = -sgn;       // branch tracker (propagators.py:1045-1047)
            c2 = det;
            // ---- this wavefront's share of step k: one writer per slot ----
#pragma unroll
            for (int i = 0; i < 5; ++i) v5[i] = wave_sum(head ? v5[i] : 0.0);
            if (lane == 0) {
                double *pp = R.partials + ((size_t)k * R.slots + slot) * 5;
#pragma unroll
                for (int i = 0; i < 5; ++i) __builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double *)(pp + i), v5[i]);
            }
        }
        // ---- the trajectory goes back
        if (active) {
            if (r < D) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int b = 0; b < D; ++b) {
                        M[(h ? DD : 0) + r * D + b] = cur[h][0][b];
                        M[(h ? DD : 0) + 2 * DD + r * D + b] = cur[h][1][b];
                    }
                qp[r] = q; qp[D + r] = p;
            }
            if (r == 0) {
                A.st.act[tr] = S; A.st.sgn[tr] = sgn;
                ((cplx *)A.st.c2)[tr] = c2;
            }
        }
    }
}

template <int D, int DP, bool DIAG, bool MODAL>
int launch_one(const RunArgs &a, int grid, hipStream_t s) {
    const size_t lds = RunLinLayout<D, DP, DIAG>::bytes;
    if (hipFuncSetAttribute((const void *)hk_run_lin_kernel<D, DP, DIAG, MODAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
        return sc_check_launch("sc_hk_run (LDS attribute)");
    hipLaunchKernelGGL((hk_run_lin_kernel<D, DP, DIAG, MODAL>), dim3(grid), dim3(256), lds, s, a);
    const int rc = sc_check_launch("sc_hk_run (constant-Hessian whole-loop kernel)");
    return rc == SC_OK ? 1 : rc;
}
template <int D, int DP, bool DIAG>
int launch(const RunArgs &a, int grid, hipStream_t s, int do_launch) {
    if (a.mode_prop) {
        if constexpr (DIAG) return 0;            // the transformed prefactor constants are dense: no modal kernel for diagonal widths
        else return do_launch ? launch_one<D, DP, DIAG, true>(a, grid, s) : 1;
    }
    return do_launch ? launch_one<D, DP, DIAG, false>(a, grid, s) : 1;
}

}  // namespace

int sc_launch_run_lin(const RunArgs &a, int grid, hipStream_t s, int do_launch) {
    const int D = a.step.st.dim, dp = a.step.hk.dprime;
    const bool diag = a.step.hk.diag != 0;
    if (!diag && !a.step.hk.real_lr) return 0;
#define SC_RUNLIN_CASE(D_, DP_, DIAG_) if (D == D_ && dp == DP_ && diag == DIAG_) return launch<D_, DP_, DIAG_>(a, grid, s, do_launch);
    SC_RUNLIN_CASE(12, 6, false) SC_RUNLIN_CASE(12, 12, true) SC_RUNLIN_CASE(9, 3, false) SC_RUNLIN_CASE(9, 9, true)
    SC_RUNLIN_CASE(6, 6, true) SC_RUNLIN_CASE(6, 6, false) SC_RUNLIN_CASE(3, 3, true)
    SC_RUNLIN_CASE(6, 1, false) SC_RUNLIN_CASE(9, 4, false) SC_RUNLIN_CASE(12, 7, false)
#undef SC_RUNLIN_CASE
    return 0;
}
