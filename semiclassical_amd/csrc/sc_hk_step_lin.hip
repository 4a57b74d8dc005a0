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
// shapes: config 3 (methylium, D = 12, d' = 6, n = 1e5) 1.78 -> 0.20 ms (docs/NOTEBOOK.md section 4.2).
#include "sc_common.h"
#include "sc_row16.h"

#ifndef SC_LIN_OCC
#define SC_LIN_OCC 2      // waves per SIMD the kernel is compiled for
#endif
#ifndef SC_LIN_FORCE_FIXUP
#define SC_LIN_FORCE_FIXUP 0     // 1: variant library that hands every determinant to the pivoted fix-up launch
#endif

#ifndef SC_LIN_DMA
#define SC_LIN_DMA 1      // 0: variant library without the LDS-DMA row prefetch (every shape on the direct loads)
#endif

namespace {

// Row prefetch through LDS (even D: a 16-byte unit never straddles two matrix rows).  The four trajectories of a wavefront
// are contiguous in memory; per HALF of the product ([Mqq, Mpq], then [Mqp, Mpp]) they are 4 D^2 16-byte units, which the
// wavefront requests with fully coalesced LDS-DMA loads (global_load_lds_dwordx4, 1 KB per instruction, no registers)
// into ITS OWN buffer: no workgroup barrier, the wavefront waits for its own vmcnt.  Lane (g, r) then reads row r of its
// trajectory from LDS.  The direct loads this replaces had every lane fetch its 8 D-byte row with 16-byte loads at a lane
// stride of 8 D bytes: 48 partially used 64-byte requests per instruction, and a memory round trip exposed in every pass.
template <int D>
struct LinDma {
    static constexpr bool on = SC_LIN_DMA && (D % 2 == 0);
    static constexpr int DD = D * D;
    static constexpr int half_units = 4 * DD;                 // 16-byte units per wavefront and half
    static constexpr int ndma = (half_units + 63) / 64;       // requests per half
    static constexpr int wave_units = ndma * 64 + 64;         // + one request of per-trajectory scalars
    static constexpr int zero_units = (DD + D + 1) / 2;       // rows of zeros for the lanes r >= D of a 16-lane row
    static constexpr int units = on ? 4 * wave_units + zero_units : 1;
};

// complex constant from LDS, read as two doubles
__device__ __forceinline__ cplx lds_cplx(const cplx *p) { const double *d = (const double *)p; return c_make(d[0], d[1]); }
__device__ __forceinline__ void lin_opaque(kptr &p) { asm volatile("" : "+s"(p)); }
__device__ __forceinline__ void lin_opaque(int &v) { asm volatile("" : "+v"(v)); }

// the way back: the lane's new rows of two blocks into the wavefront's buffer (same addresses as lin_lds_rows) ...
template <int D, int I>
__device__ __forceinline__ void lin_lds_put_rows(unsigned addr, const double (&Xq)[D], const double (&Xp)[D]) {
    if constexpr (I < D / 2) {
        const lin_d2v vq = {Xq[2 * I], Xq[2 * I + 1]}, vp = {Xp[2 * I], Xp[2 * I + 1]};
        asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(vq), "n"(16 * I) : "memory");
        asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(vp), "n"(8 * D * D + 16 * I) : "memory");
        lin_lds_put_rows<D, I + 1>(addr, Xq, Xp);
    }
}
// ... and out of it in the order of the requests: unit 64 j + lane
template <int NJ, int J>
__device__ __forceinline__ void lin_lds_unit_reads(unsigned addr, lin_d2v (&v)[NJ]) {
    if constexpr (J < NJ) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(v[J]) : "v"(addr), "n"(1024 * J));
        lin_lds_unit_reads<NJ, J + 1>(addr, v);
    }
}
template <int NJ>
__device__ __forceinline__ void lin_lds_units(unsigned addr, lin_d2v (&v)[NJ]) {
    lin_lds_unit_reads<NJ, 0>(addr, v);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(v[j]));
}

template <int D, int DP, bool DIAG>
struct LinLayout {
    // doubles: H rows [16][D], Phi rows [16][4 D], per-lane vectors [8][16]; complex rows L1, L2 [16][D] and R1, R2 [16][d']
    // (dense widths)
    // Row pitches: every lane of a 16-lane row reads ITS row of a constant, so the pitch decides the LDS banks the lanes
    // meet on.  Unpadded, Phi's rows (4 D doubles = 96 dwords at D = 12) put all lanes on one bank: a 12-way conflict on
    // every read of the product loop, 87 % of the kernel's LDS cycles (SQ_LDS_BANK_CONFLICT).  One double (one complex
    // number) more per row spreads the lanes over the banks.
    // L1, L2, R1, R2 are REAL here (round 4): they are products of U and real symmetric square roots of the positive
    // semi-definite width matrices; the reference carries them as complex128 with imaginary parts that are zero or rounding
    // dust (1e-24 for methylium).  sc_hk_consts.real_lr says so; widths with genuinely complex roots take the LDS kernel.
    static constexpr int PH = D + 1, PP = 4 * D + 1, PL = D + 1, PR = DP + 1;
    static constexpr int n_real = 16 * PH + 16 * PP + 8 * 16;
    static constexpr int n_lr = DIAG ? 0 : 2 * 16 * PL + 2 * 16 * PR;
    static constexpr size_t bytes = (size_t)n_real * 8 + (size_t)n_lr * 8 + 16 * 8;
};

// Waves per SIMD: two for every shape (256 registers per lane).  The loop of the prefetching shapes must not spill: a scratch
// reload is a vector-memory operation, and its vmcnt wait would also wait for every row request in flight (measured with
// the prefactor accumulated from whole rows: 115 spilled registers at two waves, 0.40 ms with one wave and accumulator
// registers as spill space -- no faster than without the prefetch; accumulated half by half the kernel needs 253).
template <int D> constexpr int lin_occ() { return SC_LIN_OCC; }

template <int D, int DP, bool DIAG>
__global__ __launch_bounds__(256, lin_occ<D>()) void hk_step_lin_kernel(StepArgs A) {
    typedef LinLayout<D, DP, DIAG> L;
    constexpr int W = 2 * D, DD = D * D, N = DIAG ? D : DP;
    extern __shared__ double2 smem2[];
    const int tid = threadIdx.x, r = tid & 15, grp = tid >> 4;
    const bool do_step = (A.mode & 0xff) == 0;
    const double dt = A.dt, hh = 0.5 * dt, h6 = dt / 6.0;

    double *ls = (double *)smem2;
    constexpr int PH = L::PH, PP = L::PP, PL = L::PL, PR = L::PR;
    double *sH = ls;    ls += 16 * PH;
    double *sPhi = ls;  ls += 16 * PP;             // row a: Phi_qq[a][:], Phi_qp[a][:], Phi_pq[a][:], Phi_pp[a][:]
    double *svec = ls;  ls += 8 * 16;              // x0, g0, 1/m, st, 1/st
    double *sL1 = ls, *sL2 = sL1 + 16 * PL;        // rows i < d' of Re L1, Re L2 (dense widths)
    double *sR1 = sL2 + 16 * PL, *sR2 = sR1 + 16 * PR; // rows b < D of Re R1, Re R2
    double *red = sL1 + L::n_lr;

    for (int e = tid; e < 16 * D; e += 256) {
        const int i = e / D, b = e - i * D;
        sH[i * PH + b] = i < D ? A.pot.par2[i * D + b] : 0.0;
        if (!DIAG) {
            sL1[i * PL + b] = i < DP ? A.hk.L1[2 * (i * D + b)] : 0.0;          // real parts of the interleaved complex arrays
            sL2[i * PL + b] = i < DP ? A.hk.L2[2 * (i * D + b)] : 0.0;
        }
    }
    if (!DIAG) {
        for (int e = tid; e < 16 * DP; e += 256) {
            const int i = e / DP, j = e - i * DP;
            sR1[i * PR + j] = i < D ? A.hk.R1[2 * (i * DP + j)] : 0.0;
            sR2[i * PR + j] = i < D ? A.hk.R2[2 * (i * DP + j)] : 0.0;
        }
    }
    for (int e = tid; e < 16 * 4 * D; e += 256) {
        const int i = e / (4 * D), k = e - i * 4 * D, blk = k / D, g = k - blk * D;      // blk: qq, qp, pq, pp
        const int row = (blk >> 1) * D + i, col = (blk & 1) * D + g;
        sPhi[i * PP + k] = (i < D && A.pot.lin_prop) ? A.pot.lin_prop[row * W + col] : 0.0;
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

    // ---- row prefetch (LinDma) ----
    typedef LinDma<D> Dm;
    __shared__ lin_d2v rowbuf[Dm::units];
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), gl = grp & 3;
    lin_d2v *wbuf = rowbuf + (Dm::on ? wave * Dm::wave_units : 0);
    const double *sbuf = (const double *)(wbuf + Dm::ndma * 64);         // the quad's (q, p), S, c2, sign: 8 D + 16 doubles
    // unit u = 64 j + lane of a half: trajectory g = u / DD, block (u / (DD / 2)) & 1, element pair u % (DD / 2); its
    // source, in doubles from the quad's first block of that half: 4 DD g + 2 DD blk + 2 (u % (DD / 2)) = 2 u + DD (u / (DD / 2))
    auto soff = [&](int j) { const int u = 64 * j + lane; return u < Dm::half_units ? 2 * u + DD * (u / (DD / 2)) : -1; };
    unsigned rowaddr = 0;         // LDS byte address of this lane's row in block 0 of the wavefront's buffer
    unsigned unitaddr = 0;        // ... of this lane's 16-byte unit of request 0
    if (Dm::on) {
        for (int e = tid; e < Dm::zero_units; e += 256) rowbuf[4 * Dm::wave_units + e] = (lin_d2v){0.0, 0.0};
        const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) lin_d2v *)rowbuf;
        rowaddr = r < D ? base + 16u * (unsigned)(wave * Dm::wave_units) + 8u * (unsigned)((gl * 2) * DD + r * D)
                        : base + 16u * (unsigned)(4 * Dm::wave_units);
        unitaddr = base + 16u * (unsigned)(wave * Dm::wave_units + lane);
        __syncthreads();
    }
    const int64_t n = A.st.n;
    // requests of a COMPLETE quad (tq + 4 <= n): one half of the monodromy rows; the per-trajectory scalars
    auto dma_half = [&](int64_t tq, int h) {
        const double *src = A.st.mono + tq * 4 * (int64_t)DD + (h ? DD : 0);
#pragma unroll
        for (int j = 0; j < Dm::ndma; ++j)
            if (soff(j) >= 0)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + soff(j)),
                                                 (__attribute__((address_space(3))) void *)(wbuf + 64 * j), 16, 0, 0);
    };
    auto dma_scalars = [&](int64_t tq) {
        // 16-byte units of one request: 4 D of (q, p) [4 trajectories x 2 D doubles, contiguous], 2 of S, 4 of c2, 2 of the signs
        const double *src = nullptr;
        if (lane < 4 * D) src = A.st.qp + tq * 2 * D + 2 * lane;
        else if (lane < 4 * D + 2) src = A.st.act + tq + 2 * (lane - 4 * D);
        else if (lane < 4 * D + 6) src = A.st.c2 + 2 * tq + 2 * (lane - 4 * D - 2);
        else if (lane < 4 * D + 8) src = A.st.sgn + tq + 2 * (lane - 4 * D - 6);
        if (src)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(wbuf + Dm::ndma * 64), 16, 0, 0);
    };
    auto prefetch = [&](int64_t t0) { dma_scalars(t0 + 4 * wave); dma_half(t0 + 4 * wave, 0); };

    double esum = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 16;
    // results of the PREVIOUS prefetching pass: their stores wait for the top of the next pass, so that the vmcnt(0) there
    // never waits for a young store
    cplx pend_det = c_make(0.0, 0.0);
    double pend_sgn = 0.0;                 // new branch sign, 0 = unchanged
    int64_t pend_tr = -1;
    int pend_weak = 0;
    auto commit = [&]() {
        if (pend_tr >= 0 && r == 0) {
            if (pend_weak && A.st.flags) {
                A.st.flags[pend_tr] = 1;
                atomicAdd(&A.st.flags[n], 1);
            } else {
                if (pend_sgn != 0.0) A.st.sgn[pend_tr] = pend_sgn;
                ((cplx *)A.st.c2)[pend_tr] = pend_det;
            }
        }
        pend_tr = -1;
    };

    // One pass over the workgroup's 16 trajectories.  FETCHED: all 16 exist and their rows and scalars were requested into
    // LDS by the previous pass (or the prologue); otherwise (ragged last pass, odd D, prefactor-only mode) direct loads.
    auto pass = [&](auto fetched_c, int64_t t0) {
        constexpr bool FETCHED = decltype(fetched_c)::value;
        const bool active = t0 + grp < n;
        const int64_t tr = active ? t0 + grp : n - 1;
        double *qp = A.st.qp + tr * 2 * D;
        double *M = A.st.mono + tr * 4 * (int64_t)DD;
        lin_opaque(ksi);
        int lofs = 0;
        lin_opaque(lofs);
        const double *cH = sH + lofs, *cPhi = sPhi + lofs;

        double q = 0.0, p = 0.0;
        // the read-modify-write operands of the row's first lane (action, previous determinant and branch sign) come with
        // the coordinates: the additions and the tracker at the end of the pass do not wait for memory
        double act_old = 0.0, sgn_old = 1.0;
        cplx prev = c_make(0.0, 0.0);
        if constexpr (FETCHED) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wavefront's scalars and first-half rows have landed
            const int rr = r < D ? r : 0;
            q = sbuf[gl * 2 * D + rr]; p = sbuf[gl * 2 * D + D + rr];
            act_old = sbuf[8 * D + gl]; prev = c_make(sbuf[8 * D + 4 + 2 * gl], sbuf[8 * D + 5 + 2 * gl]); sgn_old = sbuf[8 * D + 12 + gl];
            if (r >= D) { q = 0.0; p = 0.0; }
            commit();                                             // the previous pass's results: behind the reads above
        } else {
            WM_BLOCK {
                if (r < D) { q = qp[r]; p = qp[D + r]; }
                if (do_step && r == 0) { act_old = A.st.act[tr]; prev = ((const cplx *)A.st.c2)[tr]; sgn_old = A.st.sgn[tr]; }
            }
        }

        // ---- prefactor, accumulated HALF by half of the monodromy rows (columns of [Mqq; Mpq], then of [Mqp; Mpp]): a half
        // is consumed as soon as its product is there and never lives next to the other one ----
        // diagonal widths: mat_ab = 1/2 [st_a/si_b Mqq + si_b/st_a Mpp - i hbar st_a si_b Mqp + i/hbar Mpq/(st_a si_b)]  (:969-986)
        // dense widths:    X1 = Mqq R1 - i hbar Mqp R2, X2 = Mpp R2 + i/hbar Mpq R1 (row r; rows b of R1, R2 come from lane b);
        //                  mat' = 1/2 (L1 X1 + L2 X2): rows of X1, X2 from lane a, L1[i][a], L2[i][a] from LDS        (:969-994)
        cplx mat[N];
        constexpr int NS = DIAG ? 1 : DP;
        double s1[NS], s2[NS], t1[NS], t2[NS];     // Mqq R1, Mqp R2, Mpp R2, Mpq R1 (real: R1, R2 are)
        auto accum_half = [&](auto hc, const double (&Xhq)[D], const double (&Xhp)[D]) {
            constexpr int h = decltype(hc)::value;
            if constexpr (DIAG) {
                WM_BLOCK {
#pragma unroll
                    for (int b = 0; b < D; ++b) {
                        const double sib = ksi[b], isib = 1.0 / sib;
                        if (h == 0) mat[b] = r < D ? c_make(0.5 * (sta * isib * Xhq[b]), 0.5 * ((1.0 / SC_HBAR) * ista * isib * Xhp[b])) : c_make(0.0, 0.0);
                        else mat[b] = r < D ? c_make(mat[b].x + 0.5 * (ista * sib * Xhp[b]), mat[b].y + 0.5 * (-SC_HBAR * sta * sib * Xhq[b])) : c_make(0.0, 0.0);
                    }
                }
            } else {
                double rr[DP];
                double (&uq)[NS] = h ? s2 : s1, (&up)[NS] = h ? t1 : t2;
#pragma unroll
                for (int j = 0; j < DP; ++j) {
                    rr[j] = ((h ? sR2 : sR1) + lofs)[r * PR + j];
                    uq[j] = 0.0; up[j] = 0.0;
                }
                dpp_guard(rr);
                sfor<0, D>([&](auto bcn) {
                    constexpr int b = decltype(bcn)::value;
#pragma unroll
                    for (int j = 0; j < DP; ++j) { fmac_bc<b>(uq[j], rr[j], Xhq[b]); fmac_bc<b>(up[j], rr[j], Xhp[b]); }
                });
            }
        };

        if (FETCHED || do_step) {
            // ---- (q, p, S): the explicit RK4 stages (V = E0 + g.dr + 1/2 dr.H.dr - origin, grad = g + H.dr) ----
            double qs = q, ps = p, kqs = 0.0, kps = 0.0, qn = 0.0, pn = 0.0, red5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
            double hrow[D];
#pragma unroll
            for (int b = 0; b < D; ++b) hrow[b] = cH[r * PH + b];
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
                A.st.act[tr] = act_old + h6 * (red5[0] + 2.0 * red5[1] + 2.0 * red5[2] + red5[3]);
                esum += red5[4];
            }

            // ---- [X; Y] <- Phi [X; Y]: row g of the old blocks from lane g, Phi[r][g] from LDS.  Column c of the
            // result needs column c of the old blocks only: one half (Mqq, Mpq | Mqp, Mpp) at a time ----
            double Told[2][2][D], Xn[2][2][D];          // old and new rows: [half][q | p][column]
            if constexpr (FETCHED) {
                lin_lds_rows<D>(rowaddr, Told[0][0], Told[0][1]);
                dma_half(t0 + 4 * wave, 1);           // second half into the same buffer, under the first half's product
            } else {
                // BOTH halves requested together: one HBM round trip per trajectory instead of one per half of the product
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int b = 0; b < D; ++b) {
                        Told[h][0][b] = r < D ? M[(h ? DD : 0) + r * D + b] : 0.0;
                        Told[h][1][b] = r < D ? M[(h ? DD : 0) + 2 * DD + r * D + b] : 0.0;
                    }
            }
            auto product = [&](auto hc) {
                constexpr int h = decltype(hc)::value;
                double (&Tq)[D] = Told[h][0], (&Tp)[D] = Told[h][1], (&Xq)[D] = Xn[h][0], (&Xp)[D] = Xn[h][1];
#pragma unroll
                for (int b = 0; b < D; ++b) { Xq[b] = 0.0; Xp[b] = 0.0; }
                dpp_guard(Tq, Tp);
                sfor<0, D>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    const double fqq = cPhi[r * PP + g], fqp = cPhi[r * PP + D + g];
                    const double fpq = cPhi[r * PP + 2 * D + g], fpp = cPhi[r * PP + 3 * D + g];
                    // X[r][b] += Phi[r][g] * old[g][b]: the old row g comes from lane g inside the multiply-add
#pragma unroll
                    for (int b = 0; b < D; ++b) { fmac_bc<g>(Xq[b], Tq[b], fqq); fmac_bc<g>(Xp[b], Tq[b], fpq); }
#pragma unroll
                    for (int b = 0; b < D; ++b) { fmac_bc<g>(Xq[b], Tp[b], fqp); fmac_bc<g>(Xp[b], Tp[b], fpp); }
                });
            };
            auto store_half = [&](int h) {
                if constexpr (FETCHED) {
                    // through the wavefront's buffer (free between the row reads and the next request) and out in 1 KB
                    // instructions: the lanes' own rows are 8 D bytes apart, a store of theirs is 48 quarter-used 64-byte
                    // requests
                    if (r < D) lin_lds_put_rows<D, 0>(rowaddr, Xn[h][0], Xn[h][1]);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the rows are in LDS before other lanes read them
                    lin_d2v v[Dm::ndma];
                    lin_lds_units<Dm::ndma>(unitaddr, v);
                    double *dst = A.st.mono + (t0 + 4 * wave) * 4 * (int64_t)DD + (h ? DD : 0);
#pragma unroll
                    for (int j = 0; j < Dm::ndma; ++j)
                        if (soff(j) >= 0) *(lin_d2v *)(dst + soff(j)) = v[j];
                    return;
                }
                if (active && r < D) {
                    double *Oq = M + (h ? DD : 0), *Op = Oq + 2 * DD;
#pragma unroll
                    for (int b = 0; b < D; ++b) { Oq[r * D + b] = Xn[h][0][b]; Op[r * D + b] = Xn[h][1][b]; }
                }
            };
            product(std::integral_constant<int, 0>{});
            accum_half(std::integral_constant<int, 0>{}, Xn[0][0], Xn[0][1]);
            if constexpr (FETCHED) {
                // only the second half's requests are in flight here (the first half's stores come behind this wait)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lin_lds_rows<D>(rowaddr, Told[1][0], Told[1][1]);
            }
            store_half(0);
            product(std::integral_constant<int, 1>{});
            store_half(1);
            // operands of the next pass: requested here, they land under the rest of the prefactor
            if constexpr (FETCHED)
                if (t0 + stride + 16 <= n) prefetch(t0 + stride);
            accum_half(std::integral_constant<int, 1>{}, Xn[1][0], Xn[1][1]);
        } else {
            // prefactor only: rows of the stored blocks, half by half
            sfor<0, 2>([&](auto hc) {
                constexpr int h = decltype(hc)::value;
                double Xq[D], Xp[D];
                WM_BLOCK {
#pragma unroll
                    for (int b = 0; b < D; ++b) {
                        Xq[b] = r < D ? M[(h ? DD : 0) + r * D + b] : 0.0;
                        Xp[b] = r < D ? M[(h ? DD : 0) + 2 * DD + r * D + b] : 0.0;
                    }
                }
                accum_half(hc, Xq, Xp);
            });
        }
        if constexpr (!DIAG) {
            cplx X1[DP], X2[DP];
#pragma unroll
            for (int j = 0; j < DP; ++j) {
                X1[j] = c_make(s1[j], -SC_HBAR * s2[j]);                 // Mqq R1 - i hbar Mqp R2
                X2[j] = c_make(t1[j], (1.0 / SC_HBAR) * t2[j]);          // Mpp R2 + i/hbar Mpq R1
            }
#pragma unroll
            for (int j = 0; j < DP; ++j) mat[j] = c_make(0.0, 0.0);
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
        // determinant in the fixed pivot order (sc_row16.h); a weak pivot hands the trajectory to the fully pivoted
        // elimination of hk_step_kernel (fix-up launch of sc_hk_step, as on the separable fast path)
        int weak = SC_LIN_FORCE_FIXUP;
        const cplx det = det_rows_fixed_order<N>(mat, r, weak);
        double newsgn = 0.0;
        if (do_step) { if (prev.x < 0.0 && det.x < 0.0 && prev.y * det.y < 0.0) newsgn = -sgn_old; }
        else newsgn = 1.0;
        if (active) { pend_tr = tr; pend_det = det; pend_weak = weak; pend_sgn = newsgn; }
        if constexpr (!FETCHED) commit();
    };

    int64_t t0 = (int64_t)blockIdx.x * 16;
    if constexpr (Dm::on) {
        if (do_step) {
            if (t0 + 16 <= n) prefetch(t0);
            for (; t0 + 16 <= n; t0 += stride) pass(std::true_type{}, t0);
            commit();
        }
    }
    for (; t0 < n; t0 += stride) pass(std::false_type{}, t0);
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
    // persistent grid: one (prefetching shapes) or two workgroups per CU are resident; every further workgroup would stage
    // the constants (12 KB) again for one or two passes over 16 trajectories.  The kernel clears the energy partials it leaves.
    // the row prefetch moves 16-byte units: state arrays that are not 16-byte aligned go to the general kernel
    if (LinDma<D>::on && ((((uintptr_t)a.st.mono | (uintptr_t)a.st.qp | (uintptr_t)a.st.act | (uintptr_t)a.st.c2 | (uintptr_t)a.st.sgn) & 15) != 0))
        return 0;
    static_assert(!LinDma<D>::on || 4 * D + 8 <= 64, "one request holds the scalars of a quad");
    const int resident = lin_occ<D>() * 256;
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
    if (!diag && !a.hk.real_lr) return 0;       // genuinely complex L, R (indefinite rounding in the widths): the LDS kernel
#define SC_LIN_CASE(D_, DP_, DIAG_) if (D == D_ && dp == DP_ && diag == DIAG_) return launch<D_, DP_, DIAG_>(a, grid, s);
    SC_LIN_CASE(12, 6, false) SC_LIN_CASE(12, 12, true) SC_LIN_CASE(9, 3, false) SC_LIN_CASE(9, 9, true)
    SC_LIN_CASE(6, 6, true) SC_LIN_CASE(6, 6, false) SC_LIN_CASE(3, 3, true)
    // further molecular shapes (D = 3 N Cartesian coordinates, d' = D - 6, or D - 5 for a linear molecule)
    SC_LIN_CASE(6, 1, false) SC_LIN_CASE(9, 4, false) SC_LIN_CASE(12, 7, false) SC_LIN_CASE(15, 9, false)
    SC_LIN_CASE(15, 15, true)
#undef SC_LIN_CASE
    return 0;
}
