// Fused Herman-Kluk step for SEPARABLE potentials with DIAGONAL width matrices, D <= 64 -- "row-wave" layout.
//
// One 256-thread workgroup per trajectory (grid-stride).  Thread (w, lane) -- w = wave 0..3, lane 0..63 -- owns
// column b = lane of the rows a = 4 s + w, s = 0..NS-1 (NS = ceil(D/4) <= 16): a wave holds whole rows, a lane
// holds one column.  Consequences:
//   * every global access of a wave is one contiguous row of a monodromy plane (D x 8 bytes);
//   * the pivot row of elimination step k lives in ONE wave: the pivot search is a 64-lane DPP/readlane
//     reduction over all live columns (full column pivoting -- no restriction, no fallback), the scaled row is one
//     complex per lane (one ds_write / ds_read per step, one barrier);
//   * the pivot-column entry of a row is one lane of the wave that holds the row: v_readlane gives it as a
//     wave-uniform scalar, so the rank-1 update is 4 readlanes + 4 FMAs (scalar operand) per live row, with no
//     ds_bpermute and no per-thread multiplier arrays.
// Phases A (mode RK4, row propagators P_a), B (stream the monodromy blocks through P_a, build the prefactor
// matrix) and the branch tracker are as in sc_hk_step_sd.hip; references:
//   propagators.py:86-119, 313-383 (RK4 / EOM), potentials.py:63-134, 265-327, propagators.py:951-1052 (prefactor).
#include "sc_common.h"

namespace {

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}

// max over all 64 lanes, returned as a wave-uniform value
__device__ __forceinline__ int wave_max_i32(int v) {
    v = max(v, dpp_i32<0x128>(v));      // row_ror 8, 4, 2, 1: every lane of a 16-lane row holds the row max
    v = max(v, dpp_i32<0x124>(v));
    v = max(v, dpp_i32<0x122>(v));
    v = max(v, dpp_i32<0x121>(v));
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

__device__ __forceinline__ double readlane_f64(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ cplx c_inv_fast(cplx z) {
    const double x = c_abs2(z);
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return c_make(z.x * r, -z.y * r);
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// 8-byte buffer accesses: SGPR resource + ONE per-lane byte offset (column `lane` of row `w`) shared by all of a
// thread's elements; the row slot and the monodromy plane are selected by a wave-uniform SGPR offset.  Lanes beyond
// the last column carry the offset 2^31: the hardware bounds check makes their loads return zero and drops their
// stores -- no per-element 64-bit addresses, no exec-mask branches in the streaming phase.
__device__ __forceinline__ double buf_load_f64(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}
__device__ __forceinline__ void buf_store_f64(double d, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    u32x2 w;
    w.x = (unsigned)__double2loint(d); w.y = (unsigned)__double2hiint(d);
    __builtin_amdgcn_raw_buffer_store_b64(w, rs, voff, soff, 0);
}

struct PivotRecord {
    double re, im;
    int col, pad;
};

#ifndef PB_GROUP
#define PB_GROUP 4
#endif

template <int NS, int MINW>
__global__ __launch_bounds__(256, MINW) void hk_step_rw_kernel(StepArgs A) {
    __shared__ double prop[4 * 64];          // P_a = (p11, p12, p21, p22) of row a
    __shared__ double imass[64];
    __shared__ double scl[2 * 64];           // st, 1/st (row scaling)
    __shared__ cplx rowbuf[2][64];
    __shared__ PivotRecord pivrec[2];
    __shared__ int permseq[64];
    __shared__ double red[32];

    const int D = A.st.dim, DD = D * D, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool do_step = (A.mode & 0xff) == 0;
    const double dt = A.dt, hh = 0.5 * dt, h6 = dt / 6.0;
    const bool colok = lane < D;

    if (tid < 64) {
        const bool in = tid < D;
        const double st = in ? A.hk.st[tid] : 1.0;
        scl[tid] = st; scl[64 + tid] = 1.0 / st;
        imass[tid] = (in && do_step) ? A.pot.inv_mass[tid] : 1.0;
        prop[tid] = 1.0; prop[64 + tid] = 0.0; prop[128 + tid] = 0.0; prop[192 + tid] = 1.0;
    }
    // column scaling of this lane: si_b, 1/si_b
    const double sib = colok ? A.hk.si[lane] : 1.0, isib = 1.0 / sib;
    // byte offset of (row w, column lane) inside a plane; out of range for lanes beyond the last column
    const unsigned voff = colok ? (unsigned)(w * D + lane) * 8u : 0x80000000u;
    const unsigned pl1 = (unsigned)DD * 8u, pl2 = 2u * pl1, pl3 = 3u * pl1;
    __syncthreads();

    double esum = 0.0;
    for (int64_t tr = blockIdx.x; tr < A.st.n; tr += gridDim.x) {
        double *qp = A.st.qp + tr * 2 * D;
        double *M = A.st.mono + tr * 4 * (int64_t)DD;

        if (do_step) {
            // ---------------- phase A ----------------
            double red5[5] = {0, 0, 0, 0, 0};
            if (tid < D) {
                const double q = qp[tid], p = qp[D + tid], im = imass[tid];
                const double c0 = A.pot.par0[tid], c1 = A.pot.par1 ? A.pot.par1[tid] : 0.0;
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
                qp[tid] = q + h6 * (kq1 + 2.0 * kq2 + 2.0 * kq3 + kq4);
                qp[D + tid] = p + h6 * (kp1 + 2.0 * kp2 + 2.0 * kp3 + kp4);
                double u1 = 1.0, v1 = 0.0, u2 = 0.0, v2 = 1.0;
                rk4_pair(u1, v1, im, h1, h2, h3, h4, dt);
                rk4_pair(u2, v2, im, h1, h2, h3, h4, dt);
                prop[tid] = u1; prop[64 + tid] = u2; prop[128 + tid] = v1; prop[192 + tid] = v2;
            }
            block_sum<5>(red5, red);
            if (tid == 0) {
                A.st.act[tr] += h6 * (red5[0] + 2.0 * red5[1] + 2.0 * red5[2] + red5[3]);
                esum += red5[4];
            }
            __syncthreads();
        }

        // ---------------- phase B ----------------
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(M, 0, 4 * DD * 8, 0x00020000);
        cplx m[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int a = 4 * s + w;                               // wave-uniform
            const int al = a & 63;
            const unsigned srow = (unsigned)(4 * s * D) * 8u;      // first row of the slot (rows 4s .. 4s+3)
            double mqq = 0.0, mqp = 0.0, mpq = 0.0, mpp = 0.0;
            if (a < D) {
                mqq = buf_load_f64(rs, voff, srow);
                mqp = buf_load_f64(rs, voff, srow + pl1);
                mpq = buf_load_f64(rs, voff, srow + pl2);
                mpp = buf_load_f64(rs, voff, srow + pl3);
                if (do_step) {
                    const double p11 = prop[al], p12 = prop[64 + al], p21 = prop[128 + al], p22 = prop[192 + al];
                    const double nqq = fma(p12, mpq, p11 * mqq), npq = fma(p22, mpq, p21 * mqq);
                    const double nqp = fma(p12, mpp, p11 * mqp), npp = fma(p22, mpp, p21 * mqp);
                    mqq = nqq; mpq = npq; mqp = nqp; mpp = npp;
                    buf_store_f64(mqq, rs, voff, srow);
                    buf_store_f64(mqp, rs, voff, srow + pl1);
                    buf_store_f64(mpq, rs, voff, srow + pl2);
                    buf_store_f64(mpp, rs, voff, srow + pl3);
                }
            }
            const double sta = scl[al], ista = scl[64 + al];
            // lanes beyond the last column loaded zeros: their matrix entries are zero without a predicate
            m[s] = c_make(0.5 * (sta * isib * mqq + ista * sib * mpp),
                          0.5 * (-SC_HBAR * sta * sib * mqp + (1.0 / SC_HBAR) * ista * isib * mpq));
            // the scheduler would hoist the loads of all NS slots (2 x 4 x NS registers): keep groups of PB_GROUP slots
            if ((s % PB_GROUP) == PB_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
        }

        // ---------------- phase C: determinant, matrix in registers ----------------
        cplx det = c_make(1.0, 0.0);
        bool live = colok, singular = (A.mode & 0x100) != 0;      // 0x100: debug, skip the elimination
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            for (int kw = 0; kw < 4; ++kw) {
                const int k = 4 * s + kw;
                if (k >= D || singular) break;
                const int par = k & 1;
                if (w == kw) {
                    // this wave holds row k (slot s): search the live columns, scale the row, publish it
                    const cplx myinv = c_inv_fast(m[s]);
                    const int key = wave_max_i32(live ? ((__double2hiint(c_abs2(m[s])) & ~63) | (63 - lane)) : -1);
                    const int p = 63 - (key & 63);
                    const cplx piv = c_make(readlane_f64(m[s].x, p), readlane_f64(m[s].y, p));
                    const cplx inv = c_make(readlane_f64(myinv.x, p), readlane_f64(myinv.y, p));
                    const bool keep = live && lane != p;
                    const cplx r0 = c_mul(m[s], inv);
                    rowbuf[par][lane] = c_make(keep ? r0.x : 0.0, keep ? r0.y : 0.0);
                    if (lane == 0) {
                        PivotRecord rec;
                        rec.re = piv.x; rec.im = piv.y; rec.col = p; rec.pad = 0;
                        pivrec[par] = rec;
                        permseq[k] = p;
                    }
                }
                __syncthreads();
                const PivotRecord rec = pivrec[par];
                const cplx r = rowbuf[par][lane];
                if (rec.re == 0.0 && rec.im == 0.0) { singular = true; break; }
                if (tid < 64) det = c_mul(det, c_make(rec.re, rec.im));
                const int p = __builtin_amdgcn_readfirstlane(rec.col);
                live = live && lane != p;
                // rank-1 update of the rows below k; the multiplier of a row is lane p of the wave that holds it
#pragma unroll
                for (int s2 = s; s2 < NS; ++s2) {
                    if (s2 == s && w <= kw) continue;          // wave-uniform
                    const cplx c = c_make(readlane_f64(m[s2].x, p), readlane_f64(m[s2].y, p));
                    m[s2] = c_fnma(c, r, m[s2]);
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            if (singular && !(A.mode & 0x100)) {
                det = c_make(0.0, 0.0);
            } else if (!(A.mode & 0x100)) {
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
    if (tid == 0 && A.epart) A.epart[blockIdx.x] = esum;
}

}  // namespace

// launch the row-wave fast path; the caller has validated the arguments (separable potential, diag prefactor, D <= 64)
int sc_launch_step_rw(const StepArgs &a, hipStream_t s) {
    const int D = a.st.dim, ns = (D + 3) / 4, grid = sc_step_grid(a.st.n, D);
    const char *occ_env = getenv("SC_SD_OCC");      // experiment knob: waves per SIMD the NS=16 kernel is compiled for
    const int occ = occ_env ? atoi(occ_env) : 4;
    if (ns <= 4) hipLaunchKernelGGL((hk_step_rw_kernel<4, 4>), dim3(grid), dim3(256), 0, s, a);
    else if (ns <= 8) hipLaunchKernelGGL((hk_step_rw_kernel<8, 4>), dim3(grid), dim3(256), 0, s, a);
    else if (ns <= 12) hipLaunchKernelGGL((hk_step_rw_kernel<12, 4>), dim3(grid), dim3(256), 0, s, a);
    else if (occ >= 4) hipLaunchKernelGGL((hk_step_rw_kernel<16, 4>), dim3(grid), dim3(256), 0, s, a);
    else if (occ == 3) hipLaunchKernelGGL((hk_step_rw_kernel<16, 3>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((hk_step_rw_kernel<16, 2>), dim3(grid), dim3(256), 0, s, a);
    return sc_check_launch("sc_hk_step (separable/diagonal row-wave path)");
}
