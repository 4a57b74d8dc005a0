// Fast path of the fused Herman-Kluk step for SEPARABLE potentials (diagonal Hessian) with DIAGONAL width
// matrices and D <= 64 -- the anharmonic adiabatic-shift configurations of BASELINE.json.
//
// One 256-thread workgroup per trajectory, grid-stride.  Threads form a 16 x 16 grid (ti, tj); thread (ti, tj)
// owns the elements (a, b) = (16*ra + ti, 16*rb + tj), ra, rb < NR = ceil(D/16), of all four monodromy blocks
// and of the complex prefactor matrix.  A wave covers four consecutive rows x 16 consecutive columns per slot,
// i.e. every global access is a set of 128-byte row segments and the whole 4*D*D*8-byte state of the
// trajectory is read once and written once.
//
//   phase A  threads a < D: RK4 of mode a (q_a, p_a); S and <T+V> by a workgroup reduction.  With a diagonal
//            Hessian the monodromy elements (Mqq,Mpq)_ab and (Mqp,Mpp)_ab obey, for every b, the same linear
//            2x2 system with the stage Hessians h_s[a]; its RK4 step is the 2x2 matrix P_a, obtained by
//            pushing the unit vectors through the reference's stage formula.  P_a -> LDS.
//                                                               (propagators.py:86-119, 313-383; potentials.py)
//   phase B  every (a,b): (Mqq,Mpq)' = P_a (Mqq,Mpq), (Mqp,Mpp)' = P_a (Mqp,Mpp), store back, and form
//            mat_ab = 1/2[ st_a/si_b Mqq + si_b/st_a Mpp - i hbar st_a si_b Mqp + i/hbar Mpq/(st_a si_b) ]
//            in registers
//                                                                            (propagators.py:969-986)
//   phase C  c2 = det(mat) by Gaussian elimination with the matrix held in REGISTERS (NR*NR complex per
//            thread).  Rows are eliminated in natural order; the pivot COLUMN of row k is chosen by magnitude
//            among the live columns of the diagonal 16-column block (threshold pivoting: the 16 consecutive
//            lanes that own the row search it with integer keys and DPP rotations).  Because pivots stay inside
//            the diagonal block, finished row AND column blocks drop out statically.  The owner scales the row
//            by 1/pivot and publishes it through LDS (one barrier per elimination step); the pivot-column
//            entries a thread needs sit in its own 16-lane group and are fetched with ds_bpermute.
//            If the best in-block pivot is more than 16x smaller than the largest live entry of the row, the
//            trajectory is flagged (sc_state.flags) and its determinant is recomputed by the fully pivoted
//            LDS elimination of sc_hk_step.hip in the same stream (never observed for HK matrices so far; the
//            path is exercised by tests/test_hk_gpu.py::test_weak_pivot_fallback).
//            Then the sqrt branch tracker.                   (torch.det, propagators.py:999, 1006-1052)
#include "sc_common.h"

#ifdef SC_STAMPS
// diagnostic build only (tools/lu_stamps.py): per-wave cycle sums of the segments of an elimination step
__device__ unsigned long long g_stamps[4][16];
struct Stamps {
    unsigned long long acc[16];
    unsigned long long last;
    int base;
};
#define STAMP(st, i)                                                                          \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t_;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        (st).acc[(st).base + (i)] += t_ - (st).last;                                          \
        (st).last = t_;                                                                       \
    } while (0)
extern "C" int sc_debug_read_stamps(unsigned long long *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64);
}
#else
struct Stamps {};
#define STAMP(st, i) do { } while (0)
#endif

namespace {

template <int CTRL>
__device__ __forceinline__ int dpp_mov_i32(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}

// max over the 16 lanes of a DPP row (row_ror 8, 4, 2, 1); every lane of the row gets the result
__device__ __forceinline__ int row16_max_i32(int v) {
    v = max(v, dpp_mov_i32<0x128>(v));
    v = max(v, dpp_mov_i32<0x124>(v));
    v = max(v, dpp_mov_i32<0x122>(v));
    v = max(v, dpp_mov_i32<0x121>(v));
    return v;
}

__device__ __forceinline__ cplx row16_bcast(cplx v, int src) {
    return c_make(__shfl(v.x, src, 16), __shfl(v.y, src, 16));
}

// value of `v` in lane `src` (wave-uniform index) as a wave-uniform scalar
__device__ __forceinline__ double readlane_f64(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// 1/z with a hardware reciprocal refined by two Newton steps (full fp64 accuracy for normal |z|^2)
__device__ __forceinline__ cplx c_inv_fast(cplx z) {
    const double x = c_abs2(z);
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return c_make(z.x * r, -z.y * r);
}

struct PivotRecord {        // published by the owner of row k together with the scaled row
    double re, im;          // pivot value
    int col, pad;           // pivot column
};

// rank-1 update of the rows below row k = 16*KB + kt with the scaled pivot row r; pivot column = (slot KB, lane pl).
// (A wave-uniform switch over pl with DPP row_newbcast instead of ds_bpermute was measured SLOWER: the 16 code
// paths miss the instruction cache, ~900 cycles per step in tools/lu_stamps.py.)
template <int NR, int KB>
__device__ __forceinline__ void eliminate(cplx (&m)[NR][NR], const cplx (&r)[NR], int kt, int ti, int pl, Stamps &st) {
    cplx c[NR];
#pragma unroll
    for (int ra = KB; ra < NR; ++ra) c[ra] = row16_bcast(m[ra][KB], pl);
    STAMP(st, 3);
    if (ti <= kt) c[KB] = c_make(0.0, 0.0);
#pragma unroll
    for (int ra = KB; ra < NR; ++ra) {
#pragma unroll
        for (int rb = KB; rb < NR; ++rb) m[ra][rb] = c_fnma(c[ra], r[rb], m[ra][rb]);
    }
}

// all elimination steps of the diagonal block KB; returns false when a zero pivot was met.
// Within block KB only the columns of slot KB are consumed as pivots: `live` (per thread) says whether column
// (slot KB, lane tj) is still available; slots rb > KB are untouched, slots rb < KB are finished.  Padded columns
// (j >= D) hold zeros and can never win the magnitude search unless the whole row is zero (= singular).
template <int NR, int KB>
__device__ __forceinline__ bool eliminate_block(cplx (&m)[NR][NR], cplx &det, int D, cplx (*rowbuf)[64],
                                                PivotRecord *pivrec, int *permseq, int *weak, Stamps &st) {
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15, lane = tid & 63;
    const int nk = min(16, D - 16 * KB);
    bool live = 16 * KB + tj < D;
    for (int kt = 0; kt < nk; ++kt) {
        const int k = 16 * KB + kt;
        const int par = k & 1;
#ifdef SC_STAMPS
        st.base = ((tid >> 6) == (kt >> 2)) ? 0 : 8;      // owner wave of this step or not
        STAMP(st, 7);                                    // (restart the clock)
#endif
        if (ti == kt) {
            // key = upper 26 bits of |a_kj|^2 (as an integer) | (15 - tj); in-block candidates and the rest of the row
            const int blk = (__double2hiint(c_abs2(m[KB][KB])) & ~15) | (15 - tj);
            int key_blk = live ? blk : -1, key_out = -1;
#pragma unroll
            for (int rb = KB + 1; rb < NR; ++rb) key_out = max(key_out, __double2hiint(c_abs2(m[KB][rb])));
            // every lane inverts its own in-block candidate while the search runs; the winner's is used
            const cplx myinv = c_inv_fast(m[KB][KB]);
            key_blk = row16_max_i32(key_blk);
            const int pl = 15 - (key_blk & 15);
            // pivot value and inverse: lane pl of this 16-lane group, as wave-uniform scalars
            const int src = __builtin_amdgcn_readfirstlane((lane & ~15) | pl);
            const cplx piv = c_make(readlane_f64(m[KB][KB].x, src), readlane_f64(m[KB][KB].y, src));
            const cplx inv = c_make(readlane_f64(myinv.x, src), readlane_f64(myinv.y, src));
            const bool keep = live && tj != pl;
            const cplx r0 = c_mul(m[KB][KB], inv);
            rowbuf[par][16 * KB + tj] = c_make(keep ? r0.x : 0.0, keep ? r0.y : 0.0);
#pragma unroll
            for (int rb = KB + 1; rb < NR; ++rb) rowbuf[par][16 * rb + tj] = c_mul(m[KB][rb], inv);
            // |pivot|^2 more than 2^8 below some |a_kj|^2 outside the block: the pivoted fallback redoes it
            const unsigned long long any_small = __ballot((key_out & ~15) - (key_blk & ~15) > (8 << 20));
            if (tj == 0) {
                PivotRecord rec;
                rec.re = piv.x; rec.im = piv.y; rec.col = 16 * KB + pl; rec.pad = 0;
                pivrec[par] = rec;
                permseq[k] = 16 * KB + pl;
                if (any_small) *weak = 1;
            }
        }
        STAMP(st, 0);
        __syncthreads();
        STAMP(st, 1);
        const PivotRecord rec = pivrec[par];
        cplx r[NR];
#pragma unroll
        for (int rb = KB; rb < NR; ++rb) r[rb] = rowbuf[par][16 * rb + tj];
        STAMP(st, 2);
        if (rec.re == 0.0 && rec.im == 0.0) return false;
        if (tid < 64) det = c_mul(det, c_make(rec.re, rec.im));
        const int pl = rec.col & 15;
        live = live && tj != pl;
        eliminate<NR, KB>(m, r, kt, ti, pl, st);
        STAMP(st, 4);
    }
    return true;
}

template <int NR, int MINW>
__global__ __launch_bounds__(256, MINW) void hk_step_sd_kernel(StepArgs A) {
    __shared__ double prop[4 * 64];          // P_a = (p11, p12, p21, p22) of row a
    __shared__ double scl[4 * 64];           // st, 1/st, si, 1/si
    __shared__ cplx rowbuf[2][64];
    __shared__ PivotRecord pivrec[2];
    __shared__ int permseq[64];
    __shared__ int weak;

    const int D = A.st.dim, DD = D * D, tid = threadIdx.x;
    const int ti = tid >> 4, tj = tid & 15;
    const bool do_step = (A.mode & 0xff) == 0;
    if (tid < 64) {
        const bool in = tid < D;
        const double st = in ? A.hk.st[tid] : 1.0, si = in ? A.hk.si[tid] : 1.0;
        scl[tid] = st; scl[64 + tid] = 1.0 / st; scl[128 + tid] = si; scl[192 + tid] = 1.0 / si;
        prop[tid] = 1.0; prop[64 + tid] = 0.0; prop[128 + tid] = 0.0; prop[192 + tid] = 1.0;
    }
    __syncthreads();

    for (int64_t tr = blockIdx.x; tr < A.st.n; tr += gridDim.x) {
        double *M = A.st.mono + tr * 4 * (int64_t)DD;
        if (tid == 0) weak = (A.mode & 0x400) ? 1 : 0;     // 0x400: debug, force the fallback (SC_DEBUG_FORCE_FIXUP)

        if (do_step) {
            // row propagators P_a of this trajectory, computed by hk_modes_kernel ("phase A")
            const double *pr = A.st.work + tr * 4 * (int64_t)D;
            __syncthreads();
            for (int i = tid; i < 4 * D; i += 256) prop[(i / D) * 64 + (i % D)] = pr[i];
            __syncthreads();
        }

        // ---------------- phase B ----------------
        cplx m[NR][NR];
#pragma unroll
        for (int ra = 0; ra < NR; ++ra) {
            const int a = 16 * ra + ti;
            const bool rowok = a < D;
            const int al = a & 63;
            const double p11 = prop[al], p12 = prop[64 + al], p21 = prop[128 + al], p22 = prop[192 + al];
            const double sta = scl[al], ista = scl[64 + al];
            double vqq[NR], vqp[NR], vpq[NR], vpp[NR];
#pragma unroll
            for (int rb = 0; rb < NR; ++rb) {
                const int b = 16 * rb + tj;
                const bool ok = rowok && b < D;
                const int e = a * D + b;
                vqq[rb] = ok ? M[e] : 0.0;
                vqp[rb] = ok ? M[DD + e] : 0.0;
                vpq[rb] = ok ? M[2 * DD + e] : 0.0;
                vpp[rb] = ok ? M[3 * DD + e] : 0.0;
            }
#pragma unroll
            for (int rb = 0; rb < NR; ++rb) {
                const int b = 16 * rb + tj;
                const bool ok = rowok && b < D;
                const int e = a * D + b;
                double mqq = vqq[rb], mqp = vqp[rb], mpq = vpq[rb], mpp = vpp[rb];
                if (do_step) {
                    const double nqq = fma(p12, mpq, p11 * mqq), npq = fma(p22, mpq, p21 * mqq);
                    const double nqp = fma(p12, mpp, p11 * mqp), npp = fma(p22, mpp, p21 * mqp);
                    mqq = nqq; mpq = npq; mqp = nqp; mpp = npp;
                    if (ok) { M[e] = mqq; M[DD + e] = mqp; M[2 * DD + e] = mpq; M[3 * DD + e] = mpp; }
                }
                const int bl = b & 63;
                const double sib = scl[128 + bl], isib = scl[192 + bl];
                m[ra][rb] = ok ? c_make(0.5 * (sta * isib * mqq + ista * sib * mpp),
                                        0.5 * (-SC_HBAR * sta * sib * mqp + (1.0 / SC_HBAR) * ista * isib * mpq))
                               : c_make(0.0, 0.0);
            }
#ifdef SC_PHASEB_FENCE
            // one row slot of loads in flight at a time (register budget at 4 waves/SIMD)
            __asm__ volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#endif
        }

        // ---------------- phase C: determinant in registers ----------------
        cplx det = c_make(1.0, 0.0);
        bool singular = false;
        Stamps st;
#ifdef SC_STAMPS
        for (int i = 0; i < 16; ++i) st.acc[i] = 0;
        st.last = 0; st.base = 0;
#endif
        if (!(A.mode & 0x100)) {                 // 0x100: debug, skip the elimination (SC_DEBUG_SKIP_LU)
            bool ok = eliminate_block<NR, 0>(m, det, D, rowbuf, pivrec, permseq, &weak, st);
            if (NR > 1 && ok) ok = eliminate_block<NR, (NR > 1 ? 1 : 0)>(m, det, D, rowbuf, pivrec, permseq, &weak, st);
            if (NR > 2 && ok) ok = eliminate_block<NR, (NR > 2 ? 2 : 0)>(m, det, D, rowbuf, pivrec, permseq, &weak, st);
            if (NR > 3 && ok) ok = eliminate_block<NR, (NR > 3 ? 3 : 0)>(m, det, D, rowbuf, pivrec, permseq, &weak, st);
            singular = !ok;
        }
#ifdef SC_STAMPS
        if (blockIdx.x == 0 && (tid & 63) == 0)
            for (int i = 0; i < 16; ++i) g_stamps[tid >> 6][i] = st.acc[i];
#endif
        __syncthreads();
        if (tid == 0 && weak && A.st.flags && !(A.mode & 0x100)) {
            A.st.flags[tr] = 1;                  // c2 / sgn are left to the fully pivoted fallback
        } else if (tid == 0) {
            if (singular) {
                det = c_make(0.0, 0.0);
            } else {
                // sign of the column permutation k -> permseq[k]
                unsigned long long seen = 0ull;
                int transpositions = 0;
                for (int s = 0; s < D; ++s) {
                    if ((seen >> s) & 1ull) continue;
                    int len = 0, x = s;
                    while (!((seen >> x) & 1ull)) { seen |= 1ull << x; x = permseq[x]; ++len; }
                    transpositions += len - 1;
                }
                if (transpositions & 1) det = c_make(-det.x, -det.y);
            }
            cplx *c2 = (cplx *)A.st.c2;
            if (do_step) {
                const cplx prev = c2[tr];
                if (prev.x < 0.0 && det.x < 0.0 && prev.y * det.y < 0.0) A.st.sgn[tr] = -A.st.sgn[tr];
            } else {
                A.st.sgn[tr] = 1.0;
            }
            c2[tr] = det;
        }
        __syncthreads();
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

}  // namespace

// launch the fast path; the caller has validated the arguments (separable potential, diag prefactor, D <= 64)
int sc_launch_step_sd(const StepArgs &a, hipStream_t s) {
    const int D = a.st.dim, nr = (D + 15) / 16, grid = sc_step_grid(a.st.n, D);
    const char *occ_env = getenv("SC_SD_OCC");      // experiment knob: waves per SIMD the NR=4 kernel is compiled for
    const int occ = occ_env ? atoi(occ_env) : 4;
    if ((a.mode & 0xff) == 0) hipLaunchKernelGGL(hk_modes_kernel, dim3(grid), dim3(256), 0, s, a);
    switch (nr) {
        case 1: hipLaunchKernelGGL((hk_step_sd_kernel<1, 4>), dim3(grid), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((hk_step_sd_kernel<2, 4>), dim3(grid), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((hk_step_sd_kernel<3, 3>), dim3(grid), dim3(256), 0, s, a); break;
        default:
            if (occ >= 4) hipLaunchKernelGGL((hk_step_sd_kernel<4, 4>), dim3(grid), dim3(256), 0, s, a);
            else if (occ == 3) hipLaunchKernelGGL((hk_step_sd_kernel<4, 3>), dim3(grid), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((hk_step_sd_kernel<4, 2>), dim3(grid), dim3(256), 0, s, a);
            break;
    }
    return sc_check_launch("sc_hk_step (separable/diagonal fast path)");
}
