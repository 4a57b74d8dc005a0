// Fast path of the fused Herman-Kluk step for SEPARABLE potentials (diagonal Hessian) with DIAGONAL width
// matrices and D <= 64 -- the anharmonic adiabatic-shift configurations of BASELINE.json.
//
// One 256-thread workgroup per trajectory, grid-stride.  Threads form a 16 x 16 grid (ti, tj); thread (ti, tj)
// owns the elements (a, b) = (16*ra + ti, 16*rb + tj), ra, rb < NR = ceil(D/16), of all four monodromy blocks
// and of the complex prefactor matrix.  Wave w holds the thread rows ti = w, w+4, w+8, w+12 (consecutive matrix
// rows sit in DIFFERENT waves, which is what lets the elimination run as a pipeline); per slot a wave covers
// four rows x 16 consecutive columns, i.e. every global access is a set of 128-byte row segments and the whole
// 4*D*D*8-byte state of the trajectory is read once and written once.
//
//   modes    (hk_modes_kernel, one wavefront per trajectory, lane = mode) RK4 of (q_a, p_a); S and <T+V> by wave
//            reductions.  With a diagonal Hessian the monodromy elements (Mqq,Mpq)_ab and (Mqp,Mpp)_ab obey, for
//            every b, the same linear 2x2 system with the stage Hessians h_s[a]; its RK4 step is the 2x2 matrix
//            P_a, obtained by pushing the unit vectors through the reference's stage formula.  P_a -> st.work.
//                                                               (propagators.py:86-119, 313-383; potentials.py)
//   phase B  every (a,b): (Mqq,Mpq)' = P_a (Mqq,Mpq), (Mqp,Mpp)' = P_a (Mqp,Mpp), store back, and form
//            mat_ab = 1/2[ st_a/si_b Mqq + si_b/st_a Mpp - i hbar st_a si_b Mqp + i/hbar Mpq/(st_a si_b) ]
//            in registers.  Element addresses are a wave-uniform base plus ONE per-thread 32-bit offset.
//                                                                            (propagators.py:969-986)
//   phase C  c2 = det(mat) by Gaussian elimination with the matrix held in REGISTERS (NR*NR complex per
//            thread).  Rows are eliminated in natural order; the pivot COLUMN of row k is chosen by magnitude
//            among the live columns of the diagonal 16-column block (threshold pivoting: the 16 consecutive
//            lanes that own the row search it with integer keys and DPP rotations).  Because pivots stay inside
//            the diagonal block, finished row AND column blocks drop out statically.  The owner scales the row
//            by 1/pivot and publishes it through LDS; the pivot-column entries a thread needs sit in its own
//            16-lane group and are fetched with DPP row_newbcast.  Inside a block the four waves are NOT
//            barrier-coupled: consumers poll a tag in the pivot record (see eliminate_block).
//            If the best in-block pivot is more than 16x smaller than the largest live entry of the row, the
//            trajectory is flagged (sc_state.flags) and its determinant is recomputed by the fully pivoted
//            LDS elimination of sc_hk_step.hip in the same stream (never observed for HK matrices so far; the
//            path is exercised by tests/test_hk_gpu.py::test_weak_pivot_fallback).
//            Then the sqrt branch tracker.                   (torch.det, propagators.py:999, 1006-1052)
#include "sc_common.h"
#include "sc_hk_lu.h"

namespace {

// TILED: the monodromy blocks of a trajectory are stored as 16 x 16 tiles (sc_state.mono_layout = 1, see the header):
// tile (ra, rb) holds its part of Mqq, Mqp, Mpq, Mpp one after the other, each row-major inside the tile.  With the
// row mapping trow = 4 w + j a wave instruction then covers 512 contiguous bytes, the four waves one tile plane, and
// the workgroup walks the 4 D^2 doubles of the trajectory linearly -- measured 14 % more streaming bandwidth than the
// 128-byte row segments of the row-major layout (tools/micro/stream_patterns.hip).
template <int NR, int MINW, bool STEP, bool TILED>
__global__ __launch_bounds__(256, MINW) void hk_step_sd_kernel(StepArgs A) {
    __shared__ double prop[4 * 64];          // P_a = (p11, p12, p21, p22) of row a
    __shared__ double scl[4 * 64];           // st, 1/st, si, 1/si
    __shared__ cplx rowbuf[16][64];
    __shared__ PivotRecord pivrec[16];
    // per-trajectory results of the elimination, double-buffered by trajectory parity: thread 0 finishes trajectory t
    // (product of the partial determinants, branch tracker) while the other waves already stream trajectory t+1
    __shared__ cplx detbuf[2][16];           // signed partial pivot products of the 16 row groups
    __shared__ int weakbuf[2];               // bit 0: weak in-block pivot (-> pivoted fallback), bit 1: zero pivot

    const int D = A.st.dim, DD = D * D, tid = threadIdx.x;
    const int tj = tid & 15;
    const int trow = (tid >> 6) * 4 + ((tid >> 4) & 3);     // wave w holds rows 4w .. 4w+3 of every 16-row slot
    constexpr bool do_step = STEP;                   // false: prefactor and tracker initialisation only (t = 0)
    constexpr int NCL_BASE = 16 * (NR - 1);          // first column of the last column tile
    // per-thread element offsets: row-major (trow, tj) of a D x D plane; tiled: inside a 16-wide / the last tile
    const unsigned toff = TILED ? (unsigned)(trow * 16 + tj) : (unsigned)(trow * D + tj);
    const unsigned toffl = (unsigned)(trow * (D - NCL_BASE) + tj);
    if (tid < 64) {
        const bool in = tid < D;
        const double st = in ? A.hk.st[tid] : 1.0, si = in ? A.hk.si[tid] : 1.0;
        if (tid < 16) pivrec[tid].pad = 0;
        scl[tid] = st; scl[64 + tid] = 1.0 / st; scl[128 + tid] = si; scl[192 + tid] = 1.0 / si;
        prop[tid] = 1.0; prop[64 + tid] = 0.0; prop[128 + tid] = 0.0; prop[192 + tid] = 1.0;
    }
    __syncthreads();

    const bool rows_odd = row_order_is_odd(D);
    const int pslot = (tid / D) * 64 + (tid % D);  // where this thread's element of st.work goes in prop
    int seq0 = 0;                                  // tags of this workgroup's pivot records: unique per (trajectory, block)
    int par = 0;
    for (int64_t tr = blockIdx.x; tr < A.st.n; tr += gridDim.x, seq0 += 4, par ^= 1) {
        double *M = A.st.mono + tr * 4 * (int64_t)DD;
        int *weak = &weakbuf[par];
#ifdef SC_TUNING
        if (tid == 0) *weak = (A.mode & 0x400) ? 1 : 0;    // 0x400: tuning build, force the fallback (SC_DEBUG_FORCE_FIXUP)
#else
        if (tid == 0) *weak = 0;
#endif
        // row propagators P_a of this trajectory, computed by hk_modes_kernel ("phase A"): the load is issued here, the
        // values go to LDS once the first slot's loads are under way (every wave is past the previous trajectory's
        // phase B -- the elimination barriers lie in between -- so prop may be overwritten without another barrier)
        double prv = 0.0;
        if (do_step && tid < 4 * D) prv = A.st.work[tr * 4 * (int64_t)D + tid];

        // ---------------- phase B ----------------
        // LDS indices derived from til / tjl are recomputed per trajectory: hipcc otherwise hoists them out of the
        // trajectory loop, spills them, and reloads them one by one behind s_waitcnt vmcnt(0) in the middle of the stream
        int til = trow, tjl = tj;
        __asm__ volatile("" : "+v"(til), "+v"(tjl));
        cplx m[NR][NR];
#pragma unroll
        for (int ra = 0; ra < NR; ++ra) {
            const int a = 16 * ra + til;
            const bool rowok = a < D;
            const int al = a & 63;
            double vqq[NR], vqp[NR], vpq[NR], vpp[NR];
            // rows / columns of tile (ra, rb); plane = distance between the four blocks of an element
            const int nra = min(16, D - 16 * ra);
#pragma unroll
            for (int rb = 0; rb < NR; ++rb) {
                const bool ok = rowok && 16 * rb + tj < D;
                // wave-uniform element base (scalar registers) + one per-thread 32-bit offset
                const int ncb = rb == NR - 1 ? D - NCL_BASE : 16;
                const double *pe = M + __builtin_amdgcn_readfirstlane(TILED ? 4 * (16 * ra * D + nra * 16 * rb) : 16 * ra * D + 16 * rb);
                const int plane = __builtin_amdgcn_readfirstlane(TILED ? nra * ncb : DD);
                const unsigned to = (TILED && rb == NR - 1) ? toffl : toff;
                // four scalar bases: every access is "SGPR pair + the thread's 32-bit offset"
                const double *pe1 = pe + plane, *pe2 = pe1 + plane, *pe3 = pe2 + plane;
                vqq[rb] = ok ? pe[to] : 0.0;
                vqp[rb] = ok ? pe1[to] : 0.0;
                vpq[rb] = ok ? pe2[to] : 0.0;
                vpp[rb] = ok ? pe3[to] : 0.0;
            }
            if (do_step && ra == 0) {
                if (tid < 4 * D) prop[pslot] = prv;
                __syncthreads();
            }
            const double p11 = prop[al], p12 = prop[64 + al], p21 = prop[128 + al], p22 = prop[192 + al];
            const double sta = scl[al], ista = scl[64 + al];
#pragma unroll
            for (int rb = 0; rb < NR; ++rb) {
                const int b = 16 * rb + tj;
                const bool ok = rowok && b < D;
                const int ncb = rb == NR - 1 ? D - NCL_BASE : 16;
                double *pe = M + __builtin_amdgcn_readfirstlane(TILED ? 4 * (16 * ra * D + nra * 16 * rb) : 16 * ra * D + 16 * rb);
                const int plane = __builtin_amdgcn_readfirstlane(TILED ? nra * ncb : DD);
                const unsigned to = (TILED && rb == NR - 1) ? toffl : toff;
                double mqq = vqq[rb], mqp = vqp[rb], mpq = vpq[rb], mpp = vpp[rb];
                if (do_step) {
                    const double nqq = fma(p12, mpq, p11 * mqq), npq = fma(p22, mpq, p21 * mqq);
                    const double nqp = fma(p12, mpp, p11 * mqp), npp = fma(p22, mpp, p21 * mqp);
                    mqq = nqq; mpq = npq; mqp = nqp; mpp = npp;
                    double *pe1 = pe + plane, *pe2 = pe1 + plane, *pe3 = pe2 + plane;
                    if (ok) { pe[to] = mqq; pe1[to] = mqp; pe2[to] = mpq; pe3[to] = mpp; }
                }
                const int bl = (16 * rb + tjl) & 63;
                const double sib = scl[128 + bl], isib = scl[192 + bl];
                m[ra][rb] = ok ? c_make(0.5 * (sta * isib * mqq + ista * sib * mpp),
                                        0.5 * (-SC_HBAR * sta * sib * mqp + (1.0 / SC_HBAR) * ista * isib * mpq))
                               : c_make(0.0, 0.0);
            }
        }

        // ---------------- phase C: determinant in registers ----------------
        cplx det = c_make(1.0, 0.0);             // this row group's share of the product of pivots
#ifdef SC_TUNING
        const bool skip_lu = (A.mode & 0x100) != 0;      // tuning build only: phase ablation (SC_DEBUG_SKIP_LU)
#else
        constexpr bool skip_lu = false;
#endif
        if (!skip_lu) {
            auto wg_barrier = [] { __syncthreads(); };
            eliminate_block<NR, 0, 64>(m, det, D, seq0 + 1, rowbuf, pivrec, weak, tid, wg_barrier);
            if (NR > 1) eliminate_block<NR, (NR > 1 ? 1 : 0), 64>(m, det, D, seq0 + 2, rowbuf, pivrec, weak, tid, wg_barrier);
            if (NR > 2) eliminate_block<NR, (NR > 2 ? 2 : 0), 64>(m, det, D, seq0 + 3, rowbuf, pivrec, weak, tid, wg_barrier);
            if (NR > 3) eliminate_block<NR, (NR > 3 ? 3 : 0), 64>(m, det, D, seq0 + 4, rowbuf, pivrec, weak, tid, wg_barrier);
        }
        post_pivot_product(det, detbuf[par], tid);
        __syncthreads();
        // no barrier after this: the buffers of this parity are next written two trajectories on
        if (tid == 0 && (*weak & 1) && A.st.flags && !skip_lu) {
            A.st.flags[tr] = 1;                  // c2 / sgn are left to the fully pivoted fallback
            atomicAdd(&A.st.flags[A.st.n], 1);   // lets the fix-up launch return at once when nothing was flagged
        } else if (tid == 0) {
            const cplx c2new = (*weak & 2) ? c_make(0.0, 0.0) : finish_determinant(detbuf[par], rows_odd);
            cplx *c2 = (cplx *)A.st.c2;
            if (do_step) {
                const cplx prev = c2[tr];
                if (prev.x < 0.0 && c2new.x < 0.0 && prev.y * c2new.y < 0.0) A.st.sgn[tr] = -A.st.sgn[tr];
            } else {
                A.st.sgn[tr] = 1.0;
            }
            c2[tr] = c2new;
        }
    }
}

// "phase A" as its own launch: one wavefront per trajectory, lane = mode.  RK4 of (q_a, p_a) with the reference's
// stage formula, the action and <T+V> at the k4 stage by wave reductions, and the 2x2 RK4 propagator P_a of the
// monodromy rows (unit vectors pushed through the same stage formula) -> st.work[tr][4][D].
__global__ __launch_bounds__(256) void hk_modes_kernel(StepArgs A) {
    const int D = A.st.dim, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double dt = A.dt, hh = 0.5 * dt, h6 = dt / 6.0;
    __shared__ double wsum[4];
    double esum = 0.0;
    for (int64_t tr = (int64_t)blockIdx.x * 4 + wave; tr < A.st.n; tr += (int64_t)gridDim.x * 4) {
        double *qp = A.st.qp + tr * 2 * D;
        double *pr = A.st.work + tr * 4 * (int64_t)D;
        double red5[5] = {0, 0, 0, 0, 0};
        if (lane < D) {
            const double q = qp[lane], p = qp[D + lane], im = A.pot.inv_mass[lane];
            const double c0 = A.pot.par0[lane], c1 = A.pot.par1 ? A.pot.par1[lane] : 0.0;
            double v, g, h1, h2, h3, h4;
            sep_eval(A.pot.kind, c0, c1, q, v, g, h1);
            const double kq1 = p * im, kp1 = -g;
            red5[0] = 0.5 * p * p * im - v;
            const double q2 = q + hh * kq1, p2 = p + hh * kp1;
            sep_eval(A.pot.kind, c0, c1, q2, v, g, h2);
            const double kq2 = p2 * im, kp2 = -g;
            red5[1] = 0.5 * p2 * p2 * im - v;
            const double q3 = q + hh * kq2, p3 = p + hh * kp2;
            sep_eval(A.pot.kind, c0, c1, q3, v, g, h3);
            const double kq3 = p3 * im, kp3 = -g;
            red5[2] = 0.5 * p3 * p3 * im - v;
            const double q4 = q + dt * kq3, p4 = p + dt * kp3;
            sep_eval(A.pot.kind, c0, c1, q4, v, g, h4);
            const double kq4 = p4 * im, kp4 = -g;
            red5[3] = 0.5 * p4 * p4 * im - v;
            red5[4] = 0.5 * p4 * p4 * im + v;
            qp[lane] = q + h6 * (kq1 + 2.0 * kq2 + 2.0 * kq3 + kq4);
            qp[D + lane] = p + h6 * (kp1 + 2.0 * kp2 + 2.0 * kp3 + kp4);
            double u1 = 1.0, v1 = 0.0, u2 = 0.0, v2 = 1.0;
            rk4_pair(u1, v1, im, h1, h2, h3, h4, dt);
            rk4_pair(u2, v2, im, h1, h2, h3, h4, dt);
            pr[lane] = u1; pr[D + lane] = u2; pr[2 * D + lane] = v1; pr[3 * D + lane] = v2;
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) red5[i] = wave_sum(red5[i]);
        if (lane == 0) {
            A.st.act[tr] += h6 * (red5[0] + 2.0 * red5[1] + 2.0 * red5[2] + red5[3]);
            esum += red5[4];
        }
    }
    if (lane == 0) wsum[wave] = esum;
    __syncthreads();
    if (threadIdx.x == 0 && A.epart) A.epart[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}


// ---- D <= 16: one WAVEFRONT per trajectory, everything in registers, no barrier and no LDS hand-off in the elimination.
// Lane (rg, tj) = (lane >> 4, lane & 15) holds column tj of the matrix rows rg, rg + 4, rg + 8, rg + 12 (rows and
// columns beyond D are padded with the identity), for the four monodromy blocks and the prefactor matrix alike; a
// load / store instruction covers four 128-byte row segments.  The modes (RK4 of q, p, S and the row propagators P_a)
// are done by the first D lanes of the same wavefront; P_a reaches the lanes that hold row a through a per-wave LDS
// array.  Elimination, fully unrolled over the pivot row k (so its DPP row rg_k = k & 3 and register k >> 2 are
// static): the pivot column is the largest live entry of the WHOLE row (all columns sit in one 16-lane DPP row: true
// partial pivoting, no fallback needed), the scaled pivot row crosses to the other DPP rows by ds_bpermute, the
// multipliers come from lane (rg, pl) by the same 64-bit DPP row_newbcast / computed jump as in the big kernel, and the
// sign of the column permutation is accumulated from the live-column mask.
template <bool STEP>
__global__ __launch_bounds__(256) void hk_step_w16_kernel(StepArgs A) {
    __shared__ double prop[4][4][16];
    __shared__ double wsum[4];
    const int D = A.st.dim, DD = D * D, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rg = lane >> 4, tj = lane & 15;
    const double dt = A.dt, hh = 0.5 * dt, h6 = dt / 6.0;
    const bool colok = tj < D;
    const double sib = colok ? A.hk.si[tj] : 1.0, isib = 1.0 / sib;
    double sta[4], ista[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int r = rg + 4 * s;
        sta[s] = r < D ? A.hk.st[r] : 1.0;
        ista[s] = 1.0 / sta[s];
    }
    double esum = 0.0;
    for (int64_t tr = (int64_t)blockIdx.x * 4 + wave; tr < A.st.n; tr += (int64_t)gridDim.x * 4) {
        double *qp = A.st.qp + tr * 2 * D;
        double *M = A.st.mono + tr * 4 * (int64_t)DD;
        double p11[4] = {1, 1, 1, 1}, p12[4] = {0, 0, 0, 0}, p21[4] = {0, 0, 0, 0}, p22[4] = {1, 1, 1, 1};
        if (STEP) {
            double red5[5] = {0, 0, 0, 0, 0};
            if (lane < D) {
                const double q = qp[lane], p = qp[D + lane], im = A.pot.inv_mass[lane];
                const double c0 = A.pot.par0[lane], c1 = A.pot.par1 ? A.pot.par1[lane] : 0.0;
                double v, g, h1, h2, h3, h4;
                sep_eval(A.pot.kind, c0, c1, q, v, g, h1);
                const double kq1 = p * im, kp1 = -g;
                red5[0] = 0.5 * p * p * im - v;
                const double q2 = q + hh * kq1, p2 = p + hh * kp1;
                sep_eval(A.pot.kind, c0, c1, q2, v, g, h2);
                const double kq2 = p2 * im, kp2 = -g;
                red5[1] = 0.5 * p2 * p2 * im - v;
                const double q3 = q + hh * kq2, p3 = p + hh * kp2;
                sep_eval(A.pot.kind, c0, c1, q3, v, g, h3);
                const double kq3 = p3 * im, kp3 = -g;
                red5[2] = 0.5 * p3 * p3 * im - v;
                const double q4 = q + dt * kq3, p4 = p + dt * kp3;
                sep_eval(A.pot.kind, c0, c1, q4, v, g, h4);
                const double kq4 = p4 * im, kp4 = -g;
                red5[3] = 0.5 * p4 * p4 * im - v;
                red5[4] = 0.5 * p4 * p4 * im + v;
                qp[lane] = q + h6 * (kq1 + 2.0 * kq2 + 2.0 * kq3 + kq4);
                qp[D + lane] = p + h6 * (kp1 + 2.0 * kp2 + 2.0 * kp3 + kp4);
                double u1 = 1.0, v1 = 0.0, u2 = 0.0, v2 = 1.0;
                rk4_pair(u1, v1, im, h1, h2, h3, h4, dt);
                rk4_pair(u2, v2, im, h1, h2, h3, h4, dt);
                prop[wave][0][lane] = u1; prop[wave][1][lane] = u2; prop[wave][2][lane] = v1; prop[wave][3][lane] = v2;
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) red5[i] = wave_sum(red5[i]);
            if (lane == 0) {
                A.st.act[tr] += h6 * (red5[0] + 2.0 * red5[1] + 2.0 * red5[2] + red5[3]);
                esum += red5[4];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int r = rg + 4 * s;
                if (r < D) { p11[s] = prop[wave][0][r]; p12[s] = prop[wave][1][r]; p21[s] = prop[wave][2][r]; p22[s] = prop[wave][3][r]; }
            }
        }
        // ---- phase B
        cplx m[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int r = rg + 4 * s;
            const bool ok = colok && r < D;
            const int e = r * D + tj;
            double mqq = ok ? M[e] : 0.0, mqp = ok ? M[DD + e] : 0.0, mpq = ok ? M[2 * DD + e] : 0.0, mpp = ok ? M[3 * DD + e] : 0.0;
            if (STEP) {
                const double nqq = fma(p12[s], mpq, p11[s] * mqq), npq = fma(p22[s], mpq, p21[s] * mqq);
                const double nqp = fma(p12[s], mpp, p11[s] * mqp), npp = fma(p22[s], mpp, p21[s] * mqp);
                mqq = nqq; mpq = npq; mqp = nqp; mpp = npp;
                if (ok) { M[e] = mqq; M[DD + e] = mqp; M[2 * DD + e] = mpq; M[3 * DD + e] = mpp; }
            }
            m[s] = ok ? c_make(0.5 * (sta[s] * isib * mqq + ista[s] * sib * mpp),
                               0.5 * (-SC_HBAR * sta[s] * sib * mqp + (1.0 / SC_HBAR) * ista[s] * isib * mpq))
                      : c_make(r == tj ? 1.0 : 0.0, 0.0);
        }
        // ---- phase C
        cplx det = c_make(1.0, 0.0);
        bool live = colok, singular = false;
        int parity = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k < D && !singular) {
                constexpr int dummy = 0; (void)dummy;
                const int rgk = k & 3, sk = k >> 2;
                const int blk = (__double2hiint(c_abs2(m[sk])) & ~15) | (15 - tj);
                const int keymax = row16_max_i32((live && rg == rgk) ? blk : -1);
                const int kb = __builtin_amdgcn_readlane(keymax, 16 * rgk);
                const int pl = 15 - (kb & 15);
                const int src = 16 * rgk + pl;
                const cplx piv = c_make(readlane_f64(m[sk].x, src), readlane_f64(m[sk].y, src));
                if (piv.x == 0.0 && piv.y == 0.0) { singular = true; }
                else {
                    det = c_mul(det, piv);
                    const cplx inv = c_inv_fast(piv);
                    // sign of the column permutation: live columns to the left of the pivot column
                    const unsigned long long lm = __ballot(live && rg == 0);
                    parity ^= __popcll(lm & ((1ull << pl) - 1ull)) & 1;
                    // scaled pivot row at this lane's column (from DPP row rgk)
                    const cplx rowv = c_make(__shfl(m[sk].x, 16 * rgk + tj, 64), __shfl(m[sk].y, 16 * rgk + tj, 64));
                    const bool keep = live && tj != pl;
                    const cplx rs = c_mul(rowv, inv);
                    const cplx r = c_make(keep ? rs.x : 0.0, keep ? rs.y : 0.0);
                    cplx c[4];
                    column_fetch_n(pl, m[0], m[1], m[2], m[3], c[0], c[1], c[2], c[3]);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        if (rg + 4 * s <= k) c[s] = c_make(0.0, 0.0);        // rows 0..k are finished
                        m[s] = c_fnma(c[s], r, m[s]);
                    }
                    live = live && tj != pl;
                }
            }
        }
        if (lane == 0) {
            if (singular) det = c_make(0.0, 0.0);
            else if (parity) det = c_make(-det.x, -det.y);
            cplx *c2 = (cplx *)A.st.c2;
            if (STEP) {
                const cplx prev = c2[tr];
                if (prev.x < 0.0 && det.x < 0.0 && prev.y * det.y < 0.0) A.st.sgn[tr] = -A.st.sgn[tr];
            } else {
                A.st.sgn[tr] = 1.0;
            }
            c2[tr] = det;
        }
    }
    if (lane == 0) wsum[wave] = esum;
    __syncthreads();
    if (threadIdx.x == 0 && A.epart && STEP) A.epart[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

}  // namespace

#ifdef SC_TUNING
int sc_launch_step_ws(const StepArgs &a, hipStream_t s);       // tools/variants/sc_hk_step_ws.hip (measured, not adopted)
#endif

// launch the fast path; the caller has validated the arguments (separable potential, diag prefactor, D <= 64)
int sc_launch_step_sd(const StepArgs &a, hipStream_t s) {
    const int D = a.st.dim, nr = (D + 15) / 16, grid = sc_step_grid(a.st.n, D);
    int occ = 4;
    bool wave_kernel = D <= 16;
#ifdef SC_TUNING
    if (const char *occ_env = getenv("SC_SD_OCC")) occ = atoi(occ_env);   // waves per SIMD the NR=4 kernel is compiled for
    if (getenv("SC_NO_WAVE_KERNEL")) wave_kernel = false;
#endif
    if (wave_kernel) {
        // partial sums: only the first `wg` entries are written, the energy guard adds sc_step_grid() of them
        const int64_t quads = (a.st.n + 3) / 4;
        const int wg = (int)(quads < 2048 ? quads : 2048);
        if (a.epart && hipMemsetAsync(a.epart, 0, sizeof(double) * (size_t)grid, s) != hipSuccess)
            return sc_check_launch("sc_hk_step (partials)");
        if ((a.mode & 0xff) == 0) hipLaunchKernelGGL(hk_step_w16_kernel<true>, dim3(wg), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(hk_step_w16_kernel<false>, dim3(wg), dim3(256), 0, s, a);
        return sc_check_launch("sc_hk_step (one wavefront per trajectory)");
    }
    if ((a.mode & 0xff) == 0) hipLaunchKernelGGL(hk_modes_kernel, dim3(grid), dim3(256), 0, s, a);
    const bool step = (a.mode & 0xff) == 0;
    const bool tiled = a.st.mono_layout == SC_MONO_TILED16;
#ifdef SC_TUNING
    // SC_WS=1: the wave-specialised schedule (one producer + three eliminator groups per CU).  Measured on MI355X at
    // n = 1e5, D = 60: 6.71 ms against 5.98 ms of this kernel -- the eliminations are bound by VALU issue, not by their
    // latency, so giving them a CU of their own does not help.
    if (nr == 4 && step && a.mode == 0 && getenv("SC_WS")) return sc_launch_step_ws(a, s);
#endif
#define SC_LAUNCH_SD(NR_, OCC_)                                                                                         \
    do {                                                                                                                \
        if (step && tiled) hipLaunchKernelGGL((hk_step_sd_kernel<NR_, OCC_, true, true>), dim3(grid), dim3(256), 0, s, a);   \
        else if (step) hipLaunchKernelGGL((hk_step_sd_kernel<NR_, OCC_, true, false>), dim3(grid), dim3(256), 0, s, a);     \
        else if (tiled) hipLaunchKernelGGL((hk_step_sd_kernel<NR_, OCC_, false, true>), dim3(grid), dim3(256), 0, s, a);    \
        else hipLaunchKernelGGL((hk_step_sd_kernel<NR_, OCC_, false, false>), dim3(grid), dim3(256), 0, s, a);              \
    } while (0)
    switch (nr) {
        case 1: SC_LAUNCH_SD(1, 4); break;
        case 2: SC_LAUNCH_SD(2, 4); break;
        case 3: SC_LAUNCH_SD(3, 4); break;
        default:
#ifdef SC_TUNING
            if (occ == 3) { SC_LAUNCH_SD(4, 3); break; }
            if (occ < 3) { SC_LAUNCH_SD(4, 2); break; }
#endif
            (void)occ;
            SC_LAUNCH_SD(4, 4);
            break;
    }
#undef SC_LAUNCH_SD
    return sc_check_launch("sc_hk_step (separable/diagonal fast path)");
}
