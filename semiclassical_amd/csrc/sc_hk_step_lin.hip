// Fused Herman-Kluk step for a CONSTANT dense Hessian (MolecularHarmonicPotential, reference potentials.py:553-593) and
// small matrices (D <= 16): register-resident, one trajectory per 16-lane DPP row.
//
// With a constant Hessian the equations of motion of the monodromy blocks (reference propagators.py:342-357) are linear
// with constant coefficients,  d/dt [X; Y] = G [X; Y],  X = [Mqq|Mqp], Y = [Mpq|Mpp],  G = [[0, 1/m], [-H, 0]],  and one
// step of the reference's classical RK4 (propagators.py:86-119) IS the multiplication by the fixed 2D x 2D matrix
//     Phi = 1 + h G + (h G)^2 / 2 + (h G)^3 / 6 + (h G)^4 / 24
// (the four stages of RK4 applied to a linear autonomous system; differences to the staged evaluation are re-association
// only, ~1e-16).  The host builds Phi once per (potential, dt) (sc_potential.lin_prop); the kernel then does ONE product
// per step instead of four dependent stages.  (q, p, S) and the guard's <T+V> keep the explicit stages: S' = T - V is
// quadratic in the stage points.  Then the HK prefactor (diagonal widths: D x D, or the projected d' x d' matrix
// L (M R) of propagators.py:969-994), its determinant by lane-pivoted elimination (sc_row16.h) and the branch tracker.
//
// Mapping as in sc_wm_small.hip: lane a of a 16-lane row holds row a of every matrix of its trajectory; Phi and the other
// per-row constants sit in LDS (staged once per workgroup), uniform constants come from scalar registers.
// Replaces hk_step_kernel<true> (every matrix in LDS, one 64-thread workgroup per trajectory) for the instantiated
// shapes: config 3 (methylium, D = 12, d' = 6, n = 1e5) 1.78 -> see DESIGN.md.
#include "sc_common.h"
#include "sc_row16.h"

#ifndef SC_LIN_OCC
#define SC_LIN_OCC 2      // waves per SIMD the kernel is compiled for
#endif
#ifndef SC_LIN_FORCE_FIXUP
#define SC_LIN_FORCE_FIXUP 0     // 1: variant library that hands every determinant to the pivoted fix-up launch
#endif

namespace {

template <int D, int DP, bool DIAG>
struct LinLayout {
    // doubles: H rows [16][D], Phi rows [16][4 D], per-lane vectors [8][16]; complex rows L1, L2 [16][D] and R1, R2 [16][d']
    // (dense widths)
    static constexpr int n_real = 16 * D + 16 * 4 * D + 8 * 16;
    static constexpr int n_cplx = DIAG ? 0 : 2 * 16 * D + 2 * 16 * DP;
    static constexpr size_t bytes = (size_t)n_real * 8 + (size_t)n_cplx * 16 + 16 * 8;
};

template <int D, int DP, bool DIAG>
__global__ __launch_bounds__(256, SC_LIN_OCC) void hk_step_lin_kernel(StepArgs A) {
    typedef LinLayout<D, DP, DIAG> L;
    constexpr int W = 2 * D, DD = D * D, N = DIAG ? D : DP;
    extern __shared__ double2 smem2[];
    const int tid = threadIdx.x, r = tid & 15, grp = tid >> 4;
    const bool do_step = (A.mode & 0xff) == 0;
    const double dt = A.dt, hh = 0.5 * dt, h6 = dt / 6.0;

    double *ls = (double *)smem2;
    double *sH = ls;    ls += 16 * D;
    double *sPhi = ls;  ls += 16 * 4 * D;          // row a: Phi_qq[a][:], Phi_qp[a][:], Phi_pq[a][:], Phi_pp[a][:]
    double *svec = ls;  ls += 8 * 16;              // x0, g0, 1/m, st, 1/st
    cplx *sL1 = (cplx *)ls, *sL2 = sL1 + 16 * D;   // rows i < d' of L1, L2 (dense widths)
    cplx *sR1 = sL2 + 16 * D, *sR2 = sR1 + 16 * DP; // rows b < D of R1, R2
    double *red = (double *)(sL1 + L::n_cplx);

    for (int e = tid; e < 16 * D; e += 256) {
        const int i = e / D, b = e - i * D;
        sH[e] = i < D ? A.pot.par2[i * D + b] : 0.0;
        if (!DIAG) {
            sL1[e] = i < DP ? ((const cplx *)A.hk.L1)[i * D + b] : c_make(0.0, 0.0);
            sL2[e] = i < DP ? ((const cplx *)A.hk.L2)[i * D + b] : c_make(0.0, 0.0);
        }
    }
    if (!DIAG) {
        for (int e = tid; e < 16 * DP; e += 256) {
            const int i = e / DP, j = e - i * DP;
            sR1[e] = i < D ? ((const cplx *)A.hk.R1)[i * DP + j] : c_make(0.0, 0.0);
            sR2[e] = i < D ? ((const cplx *)A.hk.R2)[i * DP + j] : c_make(0.0, 0.0);
        }
    }
    for (int e = tid; e < 16 * 4 * D; e += 256) {
        const int i = e / (4 * D), k = e - i * 4 * D, blk = k / D, g = k - blk * D;      // blk: qq, qp, pq, pp
        const int row = (blk >> 1) * D + i, col = (blk & 1) * D + g;
        sPhi[e] = (i < D && A.pot.lin_prop) ? A.pot.lin_prop[row * W + col] : 0.0;
    }
    if (tid < 16) {
        const bool in = tid < D;
        svec[tid] = in ? A.pot.par0[tid] : 0.0;
        svec[16 + tid] = in ? A.pot.par1[tid] : 0.0;
        svec[32 + tid] = in ? A.pot.inv_mass[tid] : 0.0;
        const double st = (DIAG && in) ? A.hk.st[tid] : 1.0;
        svec[48 + tid] = st; svec[64 + tid] = 1.0 / st;
    }
    __syncthreads();
    const double x0 = svec[r], g0 = svec[16 + r], im = svec[32 + r], sta = svec[48 + r], ista = svec[64 + r];
    kptr ksi = (kptr)A.hk.si;

    double esum = 0.0;
    const int64_t n = A.st.n, stride = (int64_t)gridDim.x * 16;
    for (int64_t t0 = (int64_t)blockIdx.x * 16; t0 < n; t0 += stride) {
        const bool active = t0 + grp < n;
        const int64_t tr = active ? t0 + grp : n - 1;
        double *qp = A.st.qp + tr * 2 * D;
        double *M = A.st.mono + tr * 4 * (int64_t)DD;
        asm volatile("" : "+s"(ksi));
        int lofs = 0;
        asm volatile("" : "+v"(lofs));
        const double *cH = sH + lofs, *cPhi = sPhi + lofs;

        // ---- rows of the monodromy blocks: Xq = [Mqq | Mqp][r][:], Xp = [Mpq | Mpp][r][:] ----
        double Xq[W], Xp[W];
        double q = 0.0, p = 0.0;
        WM_BLOCK {
            if (r < D) { q = qp[r]; p = qp[D + r]; }
            if (!do_step) {
#pragma unroll
                for (int c = 0; c < W; ++c) { Xq[c] = 0.0; Xp[c] = 0.0; }
                if (r < D) {
#pragma unroll
                    for (int b = 0; b < D; ++b) {
                        Xq[b] = M[r * D + b]; Xq[D + b] = M[DD + r * D + b];
                        Xp[b] = M[2 * DD + r * D + b]; Xp[D + b] = M[3 * DD + r * D + b];
                    }
                }
            }
        }

        if (do_step) {
            // ---- (q, p, S): the explicit RK4 stages (V = E0 + g.dr + 1/2 dr.H.dr - origin, grad = g + H.dr) ----
            double qs = q, ps = p, kqs = 0.0, kps = 0.0, qn = 0.0, pn = 0.0, red5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
            double hrow[D];
#pragma unroll
            for (int b = 0; b < D; ++b) hrow[b] = cH[r * D + b];
            sfor<0, 4>([&](auto sc_) {
                constexpr int s = decltype(sc_)::value;
                if (s > 0) { const double c = (s == 3) ? dt : hh; qs = q + c * kqs; ps = p + c * kps; }
                double dr = r < D ? qs - x0 : 0.0;
                double hd = 0.0, hd2 = 0.0;         // two chains: consecutive multiply-adds do not wait for each other
                dpp_guard(dr);
                sfor<0, D>([&](auto bc_) {
                    constexpr int b = decltype(bc_)::value;
                    if (b & 1) fmac_bc<b>(hd2, dr, hrow[b]); else fmac_bc<b>(hd, dr, hrow[b]);
                });
                hd += hd2;
                const double v = dr * g0 + 0.5 * dr * hd;           // + scalar0 after the reduction
                const double kq = ps * im, kp = -(g0 + hd), t = 0.5 * ps * ps * im;
                red5[s] = t - v;
                if (s == 3) red5[4] = t + v;
                const double w = (s == 0 || s == 3) ? 1.0 : 2.0;
                qn += w * kq; pn += w * kp;
                kqs = kq; kps = kp;
            });
            {
                double one = 1.0;
                asm volatile("" : "+v"(one));
                double s5[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) s5[i] = 0.0;
                dpp_guard(red5);
                sfor<0, D>([&](auto kc) {             // sums over the lanes k < D: fused broadcast multiply-adds with 1.0
#pragma unroll
                    for (int i = 0; i < 5; ++i) fmac_bc<decltype(kc)::value>(s5[i], red5[i], one);
                });
#pragma unroll
                for (int i = 0; i < 5; ++i) red5[i] = s5[i];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) red5[s] -= A.pot.scalar0;
            red5[4] += A.pot.scalar0;
            if (active && r < D) { qp[r] = q + h6 * qn; qp[D + r] = p + h6 * pn; }
            if (active && r == 0) {
                A.st.act[tr] += h6 * (red5[0] + 2.0 * red5[1] + 2.0 * red5[2] + red5[3]);
                esum += red5[4];
            }

            // ---- [X; Y] <- Phi [X; Y]: row g of the old blocks from lane g, Phi[r][g] from LDS.  Column c of the
            // result needs column c of the old blocks only: one half (Mqq, Mpq | Mqp, Mpp) at a time keeps the live
            // set at three quarter-matrices ----
            // the step's old rows [Mqq | Mqp], [Mpq | Mpp]: BOTH halves requested together -- one HBM round trip per trajectory
            // instead of one per half of the product (the kernel waits for memory 60 % of its time, profiles/r3_wm_pmc.json)
            double Told[2][2][D];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int b = 0; b < D; ++b) {
                    Told[h][0][b] = r < D ? M[(h ? DD : 0) + r * D + b] : 0.0;
                    Told[h][1][b] = r < D ? M[(h ? DD : 0) + 2 * DD + r * D + b] : 0.0;
                }
            sfor<0, 2>([&](auto hc) {
                constexpr int c0 = decltype(hc)::value * D;
                double (&Tq)[D] = Told[decltype(hc)::value][0], (&Tp)[D] = Told[decltype(hc)::value][1];
#pragma unroll
                for (int b = 0; b < D; ++b) { Xq[c0 + b] = 0.0; Xp[c0 + b] = 0.0; }
                dpp_guard(Tq, Tp);
                sfor<0, D>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    const double fqq = cPhi[r * 4 * D + g], fqp = cPhi[r * 4 * D + D + g];
                    const double fpq = cPhi[r * 4 * D + 2 * D + g], fpp = cPhi[r * 4 * D + 3 * D + g];
                    // X[r][b] += Phi[r][g] * old[g][b]: the old row g comes from lane g inside the multiply-add
#pragma unroll
                    for (int b = 0; b < D; ++b) { fmac_bc<g>(Xq[c0 + b], Tq[b], fqq); fmac_bc<g>(Xp[c0 + b], Tq[b], fpq); }
#pragma unroll
                    for (int b = 0; b < D; ++b) { fmac_bc<g>(Xq[c0 + b], Tp[b], fqp); fmac_bc<g>(Xp[c0 + b], Tp[b], fpp); }
                });
                if (active && r < D) {
                    double *Oq = M + (c0 ? DD : 0), *Op = Oq + 2 * DD;
#pragma unroll
                    for (int b = 0; b < D; ++b) { Oq[r * D + b] = Xq[c0 + b]; Op[r * D + b] = Xp[c0 + b]; }
                }
            });
        }

        // ---- prefactor matrix, row i of it in lane i ----
        cplx mat[N];
        if (DIAG) {
            // mat_ab = 1/2 [st_a/si_b Mqq + si_b/st_a Mpp - i hbar st_a si_b Mqp + i/hbar Mpq/(st_a si_b)]   (:969-986)
            WM_BLOCK {
#pragma unroll
                for (int b = 0; b < D; ++b) {
                    const double sib = ksi[b], isib = 1.0 / sib;
                    mat[b] = r < D ? c_make(0.5 * (sta * isib * Xq[b] + ista * sib * Xp[D + b]),
                                            0.5 * (-SC_HBAR * sta * sib * Xq[D + b] + (1.0 / SC_HBAR) * ista * isib * Xp[b]))
                                   : c_make(0.0, 0.0);
                }
            }
        } else {
            // X1 = Mqq R1 - i hbar Mqp R2, X2 = Mpp R2 + i/hbar Mpq R1 (row r, in-lane with the uniform R1, R2);
            // mat' = 1/2 (L1 X1 + L2 X2): rows of X1, X2 from lane a, L1[i][a], L2[i][a] from LDS      (:969-994)
            cplx X1[DP], X2[DP];
            {
                // rows b of R1, R2 sit in lane b (LDS-staged per-lane rows); Xq, Xp are this lane's
                cplx r1[DP], r2[DP];
                cplx s1[DP], s2[DP], t1[DP], t2[DP];
#pragma unroll
                for (int j = 0; j < DP; ++j) {
                    r1[j] = sR1[r * DP + j + lofs]; r2[j] = sR2[r * DP + j + lofs];
                    s1[j] = c_make(0, 0); s2[j] = c_make(0, 0); t1[j] = c_make(0, 0); t2[j] = c_make(0, 0);
                }
                dpp_guard(r1, r2);
                sfor<0, D>([&](auto bcn) {
                    constexpr int b = decltype(bcn)::value;
#pragma unroll
                    for (int j = 0; j < DP; ++j) {
                        fmac_bc<b>(s1[j].x, r1[j].x, Xq[b]); fmac_bc<b>(s1[j].y, r1[j].y, Xq[b]);
                        fmac_bc<b>(s2[j].x, r2[j].x, Xq[D + b]); fmac_bc<b>(s2[j].y, r2[j].y, Xq[D + b]);
                        fmac_bc<b>(t1[j].x, r2[j].x, Xp[D + b]); fmac_bc<b>(t1[j].y, r2[j].y, Xp[D + b]);
                        fmac_bc<b>(t2[j].x, r1[j].x, Xp[b]); fmac_bc<b>(t2[j].y, r1[j].y, Xp[b]);
                    }
                });
#pragma unroll
                for (int j = 0; j < DP; ++j) {
                    X1[j] = c_add(s1[j], c_mul(c_make(0.0, -SC_HBAR), s2[j]));
                    X2[j] = c_add(t1[j], c_mul(c_make(0.0, 1.0 / SC_HBAR), t2[j]));
                }
            }
#pragma unroll
            for (int j = 0; j < DP; ++j) mat[j] = c_make(0.0, 0.0);
            dpp_guard(X1, X2);
            sfor<0, D>([&](auto ac) {
                constexpr int a = decltype(ac)::value;
                const cplx l1 = sL1[r * D + a + lofs], l2 = sL2[r * D + a + lofs];
#pragma unroll
                for (int j = 0; j < DP; ++j) cfma_bc<a>(mat[j], X1[j], l1);
#pragma unroll
                for (int j = 0; j < DP; ++j) cfma_bc<a>(mat[j], X2[j], l2);
            });
#pragma unroll
            for (int j = 0; j < DP; ++j) mat[j] = c_scale(mat[j], 0.5);
        }
        // determinant in the fixed pivot order (sc_row16.h); a weak pivot hands the trajectory to the fully pivoted
        // elimination of hk_step_kernel (fix-up launch of sc_hk_step, as on the separable fast path)
        int weak = SC_LIN_FORCE_FIXUP;
        const cplx det = det_rows_fixed_order<N>(mat, r, weak);
        if (active && r == 0 && weak && A.st.flags) {
            A.st.flags[tr] = 1;
            atomicAdd(&A.st.flags[n], 1);
        } else if (active && r == 0) {
            cplx *c2 = (cplx *)A.st.c2;
            if (do_step) {
                const cplx prev = c2[tr];
                if (prev.x < 0.0 && det.x < 0.0 && prev.y * det.y < 0.0) A.st.sgn[tr] = -A.st.sgn[tr];
            } else {
                A.st.sgn[tr] = 1.0;
            }
            c2[tr] = det;
        }
    }
    if (r == 0) red[grp] = esum;
    __syncthreads();
    if (tid == 0 && A.epart && do_step) {
        double s = 0.0;
        for (int g = 0; g < 16; ++g) s += red[g];
        A.epart[blockIdx.x] = s;
    }
    // the energy guard adds sc_step_grid(n, D) partials: workgroup 0 zeroes the ones no workgroup owns
    if (blockIdx.x == 0 && A.epart && do_step)
        for (int i = gridDim.x + tid; i < A.npart; i += 256) A.epart[i] = 0.0;
}

template <int D, int DP, bool DIAG>
int launch(const StepArgs &a, int grid, hipStream_t s) {
    const size_t lds = LinLayout<D, DP, DIAG>::bytes;
    if (hipFuncSetAttribute((const void *)hk_step_lin_kernel<D, DP, DIAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
        return sc_check_launch("sc_hk_step (LDS attribute)");
    // persistent grid: two workgroups per CU are resident (256 registers per lane); every further workgroup would stage the
    // constants (12 KB) again for one or two passes over 16 trajectories.  The kernel clears the energy partials it leaves.
    const int resident = 2 * 256;
    hipLaunchKernelGGL((hk_step_lin_kernel<D, DP, DIAG>), dim3(grid < resident ? grid : resident), dim3(256), lds, s, a);
    const int rc = sc_check_launch("sc_hk_step (constant-Hessian register kernel)");
    return rc == SC_OK ? 1 : rc;
}

}  // namespace

// returns 1 and launches if the shape (D, d', diagonal widths) is instantiated, 0 if not, < 0 on error.  The caller
// guarantees: SC_POT_HARMONIC_DENSE, pot.lin_prop built for this dt (mode 0), row-major monodromy blocks.
int sc_launch_step_lin(const StepArgs &a, int grid, hipStream_t s) {
    const int D = a.st.dim, dp = a.hk.dprime;
    const bool diag = a.hk.diag != 0;
#define SC_LIN_CASE(D_, DP_, DIAG_) if (D == D_ && dp == DP_ && diag == DIAG_) return launch<D_, DP_, DIAG_>(a, grid, s);
    SC_LIN_CASE(12, 6, false) SC_LIN_CASE(12, 12, true) SC_LIN_CASE(9, 3, false) SC_LIN_CASE(9, 9, true)
    SC_LIN_CASE(6, 6, true) SC_LIN_CASE(6, 6, false) SC_LIN_CASE(3, 3, true)
#undef SC_LIN_CASE
    return 0;
}
