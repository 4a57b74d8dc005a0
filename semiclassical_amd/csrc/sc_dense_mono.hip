// RK4 of the four monodromy blocks under DENSE, stage-dependent Hessians on the matrix cores, and the HK prefactor of
// the result -- the "batched D x D monodromy GEMM" of the dense-Hessian configurations (sGDML, SURVEY.md rows B + C4).
//
// With X = [Mqq | Mqp] and Y = [Mpq | Mpp] (D x 2D each) the equations of motion (reference propagators.py:352-362)
// are X' = W Y, Y' = -H X with W = diag(1/m): four GEMMs H_s X_s (D x D times D x 2D, 16 D^3 flop) per RK4 step
// (propagators.py:86-119).  Kernel `dense_mono_mfma_kernel<NT>` (NT = ceil(D/16) <= 4):
//   * one workgroup of 2 NT wavefronts per trajectory; a wavefront owns one 16-column tile of X and of Y for ALL rows
//     and keeps X0, Y0, the RK4 sums and the stage matrices in registers, in the accumulator layout of
//     v_mfma_f64_16x16x4_f64 (col = lane & 15, row = (lane >> 4) + 4 reg within a 16 x 16 tile);
//   * that layout IS the B-operand layout of the next product: register r of row tile t holds rows 16t + 4r .. + 3
//     across the four 16-lane groups, i.e. the k-slice 4(4t + r) -- the stage matrix never moves between products;
//   * the A operand (H_s, shared by all wavefronts) is staged in LDS, double buffered so that the next stage's
//     Hessian streams in behind the current stage's 16 NT^2 MFMAs; rows padded to 80 doubles make the one-double-
//     per-lane reads (4 rows x 16 columns) conflict free.  The LDS image is used as A[i][k] = Hs[k][i]: the stage
//     Hessians are symmetric (the sGDML kernel writes both triangles from one value), so no transpose is needed.
// FP64 MFMA peak on MI355X is 78.6 TFLOP/s = 64 cycles per 16x16x4 instruction per SIMD; at D = 64 the four products
// are 2048 MFMAs = 32.8k cycles per trajectory on one CU.
// `dense_prefactor_kernel` then forms the (projected) prefactor matrix from the new blocks and takes the determinant
// by pivoted LU in LDS, followed by the branch tracker (propagators.py:951-1052).
#include "sc_common.h"
#include "sc_prefactor.h"
#include "sc_hk_lu.h"

namespace {

struct MonoArgs {
    sc_state st;
    sc_hk_consts hk;
    const double *inv_mass;
    const double *hess;         // [n][4][D][D] stage Hessians, or [D][D] when hess_stride == 0 (constant Hessian)
    int64_t hess_stride;        // doubles between the Hessian blocks of consecutive trajectories (4 D D or 0)
    int64_t stage_stride;       // doubles between consecutive stages (D D or 0)
    double *msum;               // [n][4][D][D] RK4 sums of the monodromy blocks (only the D > 64 kernel)
    int panel;                  // columns of M R formed at a time in the prefactor kernel (LDS budget)
    double dt;
    int mode;                   // 0: after a step, 1: tracker initialisation (prefactor kernel only)
    int fixup;                  // LDS prefactor kernel: only the trajectories the register kernel flagged (weak pivots)
};

typedef double d4 __attribute__((ext_vector_type(4)));

#define HS 80                   // LDS row stride of the Hessian image (doubles)

template <int NT, int KT>          // NT = ceil(D/16) row tiles, KT = ceil(D/4) k-slices of four rows (compile time: the
                                  // product loop must be branch free or the accumulators bounce between register files)
__global__ __launch_bounds__(128 * NT, 1) void dense_mono_mfma_kernel(MonoArgs A) {
    extern __shared__ double2 smem2[];           // Hessian images [2][16 NT][HS], 1/m [64], X0/Y0 [waves][2][4 NT][64]
    constexpr int HB = 16 * NT * HS;             // doubles per image
    double *Hs0 = (double *)smem2, *wm = Hs0 + 2 * HB;
    // the step's initial matrices are only needed to form the stage inputs: parked in LDS (own slot per lane:
    // conflict free) they free registers for the products.  At NT = 4 only Y0 fits beside the Hessian images.
    constexpr bool PARK_X = NT < 4;
    constexpr int PARKED = PARK_X ? 2 : 1;
    double *Y0 = wm + 64 + (size_t)(threadIdx.x >> 6) * (PARKED * 4 * NT * 64) + (threadIdx.x & 63), *X0 = Y0 + 4 * NT * 64;
    const int D = A.st.dim, DD = D * D, tid = threadIdx.x, nth = 128 * NT;
    const int lane = tid & 63, wave = tid >> 6;
    const int pair = wave / NT, jt = wave % NT;          // pair 0: (Mqq, Mpq), pair 1: (Mqp, Mpp); column tile jt
    const int col = 16 * jt + (lane & 15), rg = lane >> 4;
    const bool colok = col < D;
    const unsigned toff = (unsigned)(rg * D + col);
    const double dt = A.dt, hh = 0.5 * dt, h6 = dt / 6.0;
    constexpr int HPT = 2 * NT;                          // Hessian elements staged per thread: (16 NT)^2 / (128 NT)

    if (tid < 64) wm[tid] = tid < D ? A.inv_mass[tid] : 0.0;
    for (int e = tid; e < 2 * HB; e += nth) Hs0[e] = 0.0;                   // padding rows / columns stay zero
    __syncthreads();

    for (int64_t tr = blockIdx.x; tr < A.st.n; tr += gridDim.x) {
        double *Mx = A.st.mono + tr * 4 * (int64_t)DD + (int64_t)pair * DD;
        double *My = Mx + 2 * (int64_t)DD;
        const double *Hg = A.hess + tr * A.hess_stride;
        // stage-1 Hessian -> LDS buffer 0 (row k of the image = row k of H, see the header)
        __syncthreads();
        for (int e = tid; e < DD; e += nth) Hs0[(e / D) * HS + (e % D)] = Hg[e];
        // this wavefront's column tile of X and Y
        double SX[NT][4], SY[NT][4], Xs[NT][4], Ys[NT][4], X0r[PARK_X ? 1 : NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * t + rg + 4 * r;
                const bool ok = colok && row < D;
                // wave-uniform row base + one per-thread 32-bit offset (keeps hipcc from hoisting 64-bit offsets)
                Xs[t][r] = ok ? (Mx + (16 * t + 4 * r) * D)[toff] : 0.0;
                Ys[t][r] = ok ? (My + (16 * t + 4 * r) * D)[toff] : 0.0;
                if (PARK_X) X0[(4 * t + r) * 64] = Xs[t][r]; else X0r[PARK_X ? 0 : t][r] = Xs[t][r];
                Y0[(4 * t + r) * 64] = Ys[t][r];
            }
#pragma unroll 1
        for (int st = 0; st < 4; ++st) {
            __syncthreads();                             // Hessian image of this stage complete
            const double *Hb = Hs0 + (st & 1) * HB;
            // next stage's Hessian: global -> registers now, registers -> LDS after the products
            double hn[HPT];
            const double *Hn = Hg + (st + 1) * A.stage_stride;
            if (st < 3) {
#pragma unroll
                for (int i = 0; i < HPT; ++i) {
                    const int e = tid + i * nth;
                    hn[i] = e < DD ? Hn[e] : 0.0;
                }
            }
            d4 acc[NT];
#pragma unroll
            for (int I = 0; I < NT; ++I) acc[I] = (d4){0.0, 0.0, 0.0, 0.0};
            // A operands one k-slice ahead of the products; the scheduling barriers keep hipcc from hoisting all
            // 4 NT^2 LDS reads of the stage to the top (that spills hundreds of bytes per lane)
            const double *arow = Hb + rg * HS + (lane & 15);
            double a_cur[NT], a_nxt[NT];
#pragma unroll
            for (int I = 0; I < NT; ++I) a_cur[I] = arow[16 * I];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                if (kt + 1 < KT) {
#pragma unroll
                    for (int I = 0; I < NT; ++I) a_nxt[I] = arow[4 * (kt + 1) * HS + 16 * I];
                }
                __builtin_amdgcn_sched_barrier(0);
                const double b = Xs[kt >> 2][kt & 3];
#pragma unroll
                for (int I = 0; I < NT; ++I)
                    acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[I], b, acc[I], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int I = 0; I < NT; ++I) a_cur[I] = a_nxt[I];
            }
            if (st < 3) {
                double *Hw = Hs0 + ((st + 1) & 1) * HB;
#pragma unroll
                for (int i = 0; i < HPT; ++i) {
                    const int e = tid + i * nth;
                    if (e < DD) Hw[(e / D) * HS + (e % D)] = hn[i];
                }
            }
            const double wgt = (st == 0 || st == 3) ? 1.0 : 2.0, c = (st == 2) ? dt : hh;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                __builtin_amdgcn_sched_barrier(0);       // one row tile of LDS operands (1/m, X0, Y0) at a time
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double kx = wm[16 * t + rg + 4 * r] * Ys[t][r], ky = -acc[t][r];
                    if (st == 0) { SX[t][r] = kx; SY[t][r] = ky; }
                    else { SX[t][r] = fma(wgt, kx, SX[t][r]); SY[t][r] = fma(wgt, ky, SY[t][r]); }
                    Xs[t][r] = fma(c, kx, PARK_X ? X0[(4 * t + r) * 64] : X0r[PARK_X ? 0 : t][r]);
                    Ys[t][r] = fma(c, ky, Y0[(4 * t + r) * 64]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * t + rg + 4 * r;
                if (colok && row < D) {
                    (Mx + (16 * t + 4 * r) * D)[toff] = fma(h6, SX[t][r], PARK_X ? X0[(4 * t + r) * 64] : X0r[PARK_X ? 0 : t][r]);
                    (My + (16 * t + 4 * r) * D)[toff] = fma(h6, SY[t][r], Y0[(4 * t + r) * 64]);
                }
            }
    }
}

// The same RK4 for 64 < D <= 96 (NT = 5, 6).  The register file of one CU no longer holds the whole step: a workgroup
// has NT wavefronts (one 16-column tile each) and does the two plane pairs one after the other; X0 / Y0 are re-read
// from the (still unmodified) state at every stage and the RK4 sums live in a global scratch of the size of the
// state (A.msum), so only the stage matrices and the accumulators stay in registers.  One Hessian image in LDS
// (rows padded to 112 doubles), staged per stage and pair.
#define HSB 112
template <int NT, int KT>
__global__ __launch_bounds__(64 * NT, 1) void dense_mono_mfma_big_kernel(MonoArgs A) {
    extern __shared__ double2 smem2[];           // Hessian image [16 NT][HSB], 1/m [128]
    constexpr int HB = 16 * NT * HSB;
    double *Hs0 = (double *)smem2, *wm = Hs0 + HB;
    const int D = A.st.dim, DD = D * D, tid = threadIdx.x, nth = 64 * NT;
    const int lane = tid & 63, jt = tid >> 6;
    const int col = 16 * jt + (lane & 15), rg = lane >> 4;
    const bool colok = col < D;
    const unsigned toff = (unsigned)(rg * D + col);
    const double dt = A.dt, hh = 0.5 * dt, h6 = dt / 6.0;
    for (int i = tid; i < 128; i += nth) wm[i] = i < D ? A.inv_mass[i] : 0.0;
    for (int e = tid; e < HB; e += nth) Hs0[e] = 0.0;
    __syncthreads();
    for (int64_t tr = blockIdx.x; tr < A.st.n; tr += gridDim.x) {
        const double *Hg = A.hess + tr * A.hess_stride;
#pragma unroll 1
        for (int pair = 0; pair < 2; ++pair) {
            double *Mx = A.st.mono + tr * 4 * (int64_t)DD + (int64_t)pair * DD, *My = Mx + 2 * (int64_t)DD;
            double *Sx = A.msum + tr * 4 * (int64_t)DD + (int64_t)pair * DD, *Sy = Sx + 2 * (int64_t)DD;
            double Xs[NT][4], Ys[NT][4];
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = colok && 16 * t + rg + 4 * r < D;
                    Xs[t][r] = ok ? (Mx + (16 * t + 4 * r) * D)[toff] : 0.0;
                    Ys[t][r] = ok ? (My + (16 * t + 4 * r) * D)[toff] : 0.0;
                }
#pragma unroll 1
            for (int st = 0; st < 4; ++st) {
                __syncthreads();                         // everybody is done with the previous image
                const double *Hst = Hg + st * A.stage_stride;
                for (int e = tid; e < DD; e += nth) Hs0[(e / D) * HSB + (e % D)] = Hst[e];
                __syncthreads();
                d4 acc[NT];
#pragma unroll
                for (int I = 0; I < NT; ++I) acc[I] = (d4){0.0, 0.0, 0.0, 0.0};
                const double *arow = Hs0 + rg * HSB + (lane & 15);
                double a_cur[NT], a_nxt[NT];
#pragma unroll
                for (int I = 0; I < NT; ++I) a_cur[I] = arow[16 * I];
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    if (kt + 1 < KT) {
#pragma unroll
                        for (int I = 0; I < NT; ++I) a_nxt[I] = arow[4 * (kt + 1) * HSB + 16 * I];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const double b = Xs[kt >> 2][kt & 3];
#pragma unroll
                    for (int I = 0; I < NT; ++I)
                        acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[I], b, acc[I], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int I = 0; I < NT; ++I) a_cur[I] = a_nxt[I];
                }
                const double wgt = (st == 0 || st == 3) ? 1.0 : 2.0, c = (st == 2) ? dt : hh;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = colok && 16 * t + rg + 4 * r < D;
                        if (ok) {
                            // wave-uniform row bases + one per-thread 32-bit offset (no hoisted 64-bit offsets)
                            const int rowoff = (16 * t + 4 * r) * D;
                            double *mx = Mx + rowoff, *my = My + rowoff, *sxp = Sx + rowoff, *syp = Sy + rowoff;
                            const double kx = wm[16 * t + rg + 4 * r] * Ys[t][r], ky = -acc[t][r];
                            const double x0 = mx[toff], y0 = my[toff];
                            const double sx = (st == 0 ? 0.0 : sxp[toff]) + wgt * kx, sy = (st == 0 ? 0.0 : syp[toff]) + wgt * ky;
                            if (st < 3) {
                                sxp[toff] = sx; syp[toff] = sy;
                                Xs[t][r] = fma(c, kx, x0);
                                Ys[t][r] = fma(c, ky, y0);
                            } else {
                                mx[toff] = fma(h6, sx, x0);
                                my[toff] = fma(h6, sy, y0);
                            }
                        }
                    }
                }
            }
        }
    }
}

// 64 < D <= 96 with the whole RK4 step in registers (round 2; replaces the global-scratch kernel above as the default).
// The equations of motion couple the ROWS of a monodromy block through H but not its columns: a 16-column tile of
// X = [Mqq|Mqp] together with the same tile of Y = [Mpq|Mpp] is an independent problem.  So a trajectory is cut into
// slabs of four column tiles; a 256-thread workgroup takes one slab, every wavefront one tile for ALL rows, and keeps
// X0, Y0, the stage matrices and the RK4 sums of its tile in registers (24 NT doubles + accumulators: the kernel is
// compiled for one wave per SIMD = 512 registers).  No scratch traffic at all: per trajectory the state is read and
// written once (2 x 4 D^2 x 8 B) and the four stage Hessians are read once per slab (2 NT / 4 slabs), against ~3 MB
// through the global scratch of dense_mono_mfma_big_kernel.  One Hessian image in LDS (rows padded to 112 doubles); the
// next stage's image is fetched into registers behind the current stage's MFMAs.
// DMA (round 3): the stage Hessians go from L2 / HBM straight into the LDS image with 16-byte LDS-DMA loads
// (global_load_lds_dwordx4: no registers, no index arithmetic per element -- the register path spent ~2400 instructions per
// stage, divisions by D among them, behind the 138 MFMAs of a stage, and the matrix pipe sat idle for two thirds of the
// launch).  A wave instruction fills 1 KB of the image's linear space; the lanes that fall on the padding columns are
// switched off, so padding rows and columns keep the zeros they were given at kernel start.  Needs 16-byte aligned rows
// (D even); odd D keeps the register path.
template <int NT, int KT, bool DMA>
__device__ __forceinline__ void dense_mono_mfma_slab_body(const MonoArgs &A) {
    extern __shared__ double2 smem2[];           // Hessian image [16 NT][HSB], 1/m [128]
    constexpr int HB = 16 * NT * HSB, NSLAB = (2 * NT + 3) / 4;
    constexpr int HPT = (16 * NT * 16 * NT + 255) / 256;      // Hessian elements fetched per thread (register path)
    double *Hs0 = (double *)smem2, *wm = Hs0 + HB;
    const int D = A.st.dim, DD = D * D, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, rg = lane >> 4;
    const double dt = A.dt, hh = 0.5 * dt, h6 = dt / 6.0;
    for (int i = tid; i < 128; i += 256) wm[i] = i < D ? A.inv_mass[i] : 0.0;
    for (int e = tid; e < HB; e += 256) Hs0[e] = 0.0;                        // padding rows / columns stay zero
    __syncthreads();
    // image rows 0 .. D-1 in 16-byte units: unit u = row * (HSB / 2) + cu holds columns 2 cu, 2 cu + 1
    auto fetch_image = [&](const double *src) {
        constexpr int UPR = HSB / 2;
        const int upd = D >> 1, total = D * UPR;
        for (int u0 = 64 * __builtin_amdgcn_readfirstlane(wave); u0 < total; u0 += 256) {
            const int u = u0 + lane, row = u / UPR, cu = u - row * UPR;
            if (u < total && cu < upd)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)row * D + 2 * cu),
                                                 (__attribute__((address_space(3))) void *)(Hs0 + 2 * u0), 16, 0, 0);
        }
    };
    const int64_t items = A.st.n * NSLAB;
    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int64_t tr = item / NSLAB;
        const int ct = 4 * (int)(item - tr * NSLAB) + wave;                  // column tile of [X | X'] this wavefront owns
        const bool active = ct < 2 * NT;
        const int pair = active ? ct / NT : 0, jt = active ? ct % NT : 0;    // pair 0: (Mqq, Mpq), pair 1: (Mqp, Mpp)
        const int col = 16 * jt + (lane & 15);
        const bool colok = active && col < D;
        const unsigned toff = (unsigned)(rg * D + col);
        double *Mx = A.st.mono + tr * 4 * (int64_t)DD + (int64_t)pair * DD, *My = Mx + 2 * (int64_t)DD;
        const double *Hg = A.hess + tr * A.hess_stride;
        __syncthreads();                                                     // the previous item's last image is done with
        if (DMA) fetch_image(Hg);
        else for (int e = tid; e < DD; e += 256) Hs0[(e / D) * HSB + (e % D)] = Hg[e];
        double SX[NT][4], SY[NT][4], Xs[NT][4], Ys[NT][4], X0[NT][4], Y0[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = colok && 16 * t + rg + 4 * r < D;
                X0[t][r] = Xs[t][r] = ok ? (Mx + (16 * t + 4 * r) * D)[toff] : 0.0;
                Y0[t][r] = Ys[t][r] = ok ? (My + (16 * t + 4 * r) * D)[toff] : 0.0;
                SX[t][r] = 0.0; SY[t][r] = 0.0;
            }
#pragma unroll 1
        for (int st = 0; st < 4; ++st) {
            if (DMA) __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0): this wavefront's part of the image has landed
            __syncthreads();                                                 // image of this stage complete
            double hn[DMA ? 1 : HPT];
            const double *Hn = Hg + (st + 1) * A.stage_stride;
            if (!DMA && st < 3) {
#pragma unroll
                for (int i = 0; i < HPT; ++i) {
                    const int e = tid + i * 256;
                    hn[DMA ? 0 : i] = e < DD ? Hn[e] : 0.0;
                }
            }
            d4 acc[NT];
#pragma unroll
            for (int I = 0; I < NT; ++I) acc[I] = (d4){0.0, 0.0, 0.0, 0.0};
            const double *arow = Hs0 + rg * HSB + (lane & 15);
            double a_cur[NT], a_nxt[NT];
#pragma unroll
            for (int I = 0; I < NT; ++I) a_cur[I] = arow[16 * I];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                if (kt + 1 < KT) {
#pragma unroll
                    for (int I = 0; I < NT; ++I) a_nxt[I] = arow[4 * (kt + 1) * HSB + 16 * I];
                }
                __builtin_amdgcn_sched_barrier(0);
                const double b = Xs[kt >> 2][kt & 3];
#pragma unroll
                for (int I = 0; I < NT; ++I)
                    acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[I], b, acc[I], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int I = 0; I < NT; ++I) a_cur[I] = a_nxt[I];
            }
            if (DMA && st < 3) {
                __syncthreads();                                             // everybody has read this stage's image:
                fetch_image(Hn);                                             // the next one streams in under the update below
            }
            const double wgt = (st == 0 || st == 3) ? 1.0 : 2.0, c = (st == 2) ? dt : hh;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double kx = wm[16 * t + rg + 4 * r] * Ys[t][r], ky = -acc[t][r];
                    SX[t][r] = fma(wgt, kx, SX[t][r]); SY[t][r] = fma(wgt, ky, SY[t][r]);
                    Xs[t][r] = fma(c, kx, X0[t][r]);
                    Ys[t][r] = fma(c, ky, Y0[t][r]);
                }
            if (!DMA && st < 3) {
                __syncthreads();                                             // everybody has read this stage's image
#pragma unroll
                for (int i = 0; i < HPT; ++i) {
                    const int e = tid + i * 256;
                    if (e < DD) Hs0[(e / D) * HSB + (e % D)] = hn[DMA ? 0 : i];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (colok && 16 * t + rg + 4 * r < D) {
                    (Mx + (16 * t + 4 * r) * D)[toff] = fma(h6, SX[t][r], X0[t][r]);
                    (My + (16 * t + 4 * r) * D)[toff] = fma(h6, SY[t][r], Y0[t][r]);
                }
            }
    }
}

// TWO images (D <= 90: (2 D + 2) rows of 112 doubles fit the 160 KB), stages as ONE stream across the slabs a workgroup
// processes: while the products of stage g run on image g & 1, the Hessian of stage g + 1 -- the next stage of this slab or
// stage 0 of the workgroup's next slab -- is requested into the other image, two LDS-DMA loads per k-slice, from offsets
// computed once per kernel.  Their issue slots lie in the shadow of the MFMAs (a k-slice keeps the matrix pipe busy for
// ~400 cycles and needs three LDS reads), they have more than half a stage to land, and a stage needs ONE barrier: with
// one image the request went out behind the products, cost ~25 instructions per load in the open and its HBM round trip
// (the 2.6 GB of stage Hessians of a batch stay in no cache) lay bare in front of the next stage.
// The images are packed (image 1 starts at row D): the k-slice that reaches beyond row D - 1 reads rows of the other image /
// of a zeroed tail, multiplied by rows of the stage matrix that are exactly zero.  1/m lives in registers.
// hipcc must not know that the A operands are LDS reads: behind an LDS-DMA it puts `s_waitcnt vmcnt(0)` in front of every
// LDS read that may alias the DMA's destination -- and in front of every __syncthreads() (a workgroup release fence).  So
// the operand reads are inline-assembly ds_read2_b64 with their own lgkmcnt waits, and the barrier inside the loop is a
// bare s_barrier behind an explicit wait.
typedef double sc_d2v __attribute__((ext_vector_type(2)));
// A operands of one k-slice: doubles 0, 16, 32, ... (16 (NT-1)) behind byte address `addr` of the LDS image, as NT/2
// ds_read2_b64 (offsets in units of 8 bytes) the compiler does not see as LDS reads
template <int NT>
__device__ __forceinline__ void lds_a_operands(unsigned addr, sc_d2v (&a)[3]) {
    static_assert(NT == 5 || NT == 6, "slab kernels: 64 < D <= 96");
    if constexpr (NT == 6)
        asm volatile("ds_read2_b64 %0, %3 offset1:16\n\tds_read2_b64 %1, %3 offset0:32 offset1:48\n\tds_read2_b64 %2, %3 offset0:64 offset1:80"
                     : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]) : "v"(addr));
    else
        asm volatile("ds_read2_b64 %0, %3 offset1:16\n\tds_read2_b64 %1, %3 offset0:32 offset1:48\n\tds_read2_b64 %2, %3 offset0:64 offset1:64"
                     : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]) : "v"(addr));
}
// wait until at most the LDS reads of ONE k-slice (three instructions) are outstanding; the operands pass through the
// statement so that their consumers are ordered behind it
__device__ __forceinline__ void lds_a_wait_previous(sc_d2v (&a)[3]) {
    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]));
}
__device__ __forceinline__ void lds_a_wait_all(sc_d2v (&a)[3]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]));
}

// -DGDML_PHASE_CLOCK (variant library, tools/mono_phases.py): wave 0 of workgroup 0 accumulates the shader cycles of the phases
#ifdef GDML_PHASE_CLOCK
__device__ unsigned long long *g_mono_clock = nullptr;
#define MONO_TICK(slot) do { if (pc_on) { const unsigned long long now_ = clock64(); pc[slot] += now_ - pc_last; pc_last = now_; } } while (0)
#else
#define MONO_TICK(slot) do { } while (0)
#endif

template <int NT, int KT>
__global__ __launch_bounds__(256, 1) void dense_mono_mfma_slab_dma2(MonoArgs A) {
#ifdef GDML_PHASE_CLOCK
    const bool pc_on = blockIdx.x == 0 && threadIdx.x == 0 && g_mono_clock != nullptr;
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pc_last = clock64();
#endif
    extern __shared__ double2 smem2[];           // Hessian images [2 D + 2][HSB]
    constexpr int NSLAB = (2 * NT + 3) / 4, UPR = HSB / 2;
    constexpr int NDMA = (16 * NT * UPR + 255) / 256;          // LDS-DMA instructions per image and wavefront
    static_assert(2 * KT >= NDMA + 2, "two requests per k-slice have to fit into a stage's product loop");
    double *Hs0 = (double *)smem2;
    const int D = A.st.dim, DD = D * D, tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), rg = lane >> 4;
    const double dt = A.dt, hh = 0.5 * dt, h6 = dt / 6.0;
    for (int e = tid; e < (2 * D + 2) * HSB; e += 256) Hs0[e] = 0.0;        // padding columns and the tail rows stay zero
    double wmr[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) wmr[t][r] = 16 * t + rg + 4 * r < D ? A.inv_mass[16 * t + rg + 4 * r] : 0.0;
    // image rows 0 .. D-1 in 16-byte units: unit u = row * UPR + cu holds columns 2 cu, 2 cu + 1.  Request j of a wavefront
    // covers the 64 units from u0_j = min(64 wave + 256 j, total - 64) (the last requests re-cover the image's tail, so that
    // every wavefront issues exactly NDMA of them); lanes on padding columns are switched off (offset -1).
    const int upd = D >> 1, total = D * UPR;
    int soff[NDMA];                                                         // byte offset of this lane's unit inside a D x D Hessian
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
        const int u = min(64 * wave + 256 * j, total - 64) + lane, row = u / UPR, cu = u - row * UPR;
        soff[j] = cu < upd ? 8 * (row * D + 2 * cu) : -1;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)Hs0;
    __syncthreads();
    auto request = [&](const double *src, int buf, int j) {                 // request j of the image of `src` into image `buf`
        const int u0 = min(64 * wave + 256 * j, total - 64);
        if (soff[j] >= 0)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)src + (unsigned)soff[j]),
                                             (__attribute__((address_space(3))) void *)(Hs0 + (size_t)buf * D * HSB + 2 * u0), 16, 0, 0);
    };
    // Work items = (trajectory, slab).  The NSLAB slabs of a trajectory read the SAME four Hessian images (NSLAB x 4 x 8 D^2
    // bytes of the launch's traffic): workgroup b lands on XCD b % 8 (round-robin dispatch), so the slabs of a trajectory are
    // dealt to workgroups of ONE XCD -- b = 8 q + xcd takes trajectory 8 (q / NSLAB) + xcd, slab q % NSLAB -- and meet in its L2
    // instead of fetching the images NSLAB times over the fabric (FETCH_SIZE 11.4 GB per launch at D = 90, n = 1e4, against
    // 5.2 GB algorithmic).  Needs a grid that is a multiple of 8; the virtual item range is rounded up, items beyond n skipped.
    const bool by_xcd = (gridDim.x & 7) == 0;
    const int64_t n_round = by_xcd ? (A.st.n + 7) / 8 * 8 : A.st.n, items = n_round * NSLAB;
    auto trajectory_of = [&](int64_t item, int &slab) -> int64_t {
        if (!by_xcd) { slab = (int)(item % NSLAB); return item / NSLAB; }
        const int64_t q = item >> 3;
        slab = (int)(q % NSLAB);
        return (q / NSLAB) * 8 + (item & 7);
    };
    // first item of this workgroup with a trajectory (the virtual range has holes beyond n)
    int64_t first = blockIdx.x;
    { int sl; while (first < items && trajectory_of(first, sl) >= A.st.n) first += gridDim.x; }
    if (first < items) {
        int sl;
        const double *H0 = A.hess + trajectory_of(first, sl) * A.hess_stride;
#pragma unroll
        for (int j = 0; j < NDMA; ++j) request(H0, 0, j);                   // stage 0 of the first slab (the only unhidden image)
    }
    int g = 0;                                                               // stages this workgroup has started: image g & 1
    for (int64_t item = first; item < items; ) {
        int slab;
        const int64_t tr = trajectory_of(item, slab);
        const int ct = 4 * slab + wave;                                      // column tile of [X | X'] this wavefront owns
        const bool active = ct < 2 * NT;
        const int pair = active ? ct / NT : 0, jt = active ? ct % NT : 0;    // pair 0: (Mqq, Mpq), pair 1: (Mqp, Mpp)
        const int col = 16 * jt + (lane & 15);
        const bool colok = active && col < D;
        const unsigned toff = (unsigned)(rg * D + col);
        double *Mx = A.st.mono + tr * 4 * (int64_t)DD + (int64_t)pair * DD, *My = Mx + 2 * (int64_t)DD;
        const double *Hg = A.hess + tr * A.hess_stride;
        int64_t nitem = item + gridDim.x;
        int nslab;
        while (nitem < items && trajectory_of(nitem, nslab) >= A.st.n) nitem += gridDim.x;
        const double *Hnext_item = nitem < items ? A.hess + trajectory_of(nitem, nslab) * A.hess_stride : nullptr;
        double SX[NT][4], SY[NT][4], Xs[NT][4], Ys[NT][4], X0[NT][4], Y0[NT][4];
        MONO_TICK(0);                                                        // item bookkeeping
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = colok && 16 * t + rg + 4 * r < D;
                X0[t][r] = Xs[t][r] = ok ? (Mx + (16 * t + 4 * r) * D)[toff] : 0.0;
                Y0[t][r] = Ys[t][r] = ok ? (My + (16 * t + 4 * r) * D)[toff] : 0.0;
                SX[t][r] = 0.0; SY[t][r] = 0.0;
            }
#pragma unroll 1
        for (int st = 0; st < 4; ++st, ++g) {
            MONO_TICK(st == 0 ? 1 : 4);                                      // 1: loads of the tile's rows issued; 4: RK4 update
            // the image of this stage was requested during the previous stage's products: this wavefront's part has landed ...
            __builtin_amdgcn_s_waitcnt(0x0F70);
            // ... and everybody's; also: every wavefront is through the products of the previous stage, whose image the
            // requests below overwrite
            __builtin_amdgcn_s_barrier();
            const double *Hnext = st < 3 ? Hg + (st + 1) * A.stage_stride : Hnext_item;
            const int nbuf = (g + 1) & 1;
            MONO_TICK(2);                                                    // wait for the image (and, stage 0, the rows) + barrier
            d4 acc[NT];
#pragma unroll
            for (int I = 0; I < NT; ++I) acc[I] = (d4){0.0, 0.0, 0.0, 0.0};
            // byte address of this lane's first A operand in the image of this stage; + 4 rows per k-slice
            unsigned aaddr = lds_base + 8u * (unsigned)((g & 1) * D * HSB + rg * HSB + (lane & 15));
            sc_d2v a_cur[3], a_nxt[3];
            lds_a_operands<NT>(aaddr, a_cur);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                if (kt + 1 < KT) {
                    aaddr += 8u * 4u * HSB;
                    lds_a_operands<NT>(aaddr, a_nxt);
                    lds_a_wait_previous(a_cur);
                } else {
                    lds_a_wait_all(a_cur);
                }
                const double b = Xs[kt >> 2][kt & 3];
#pragma unroll
                for (int I = 0; I < NT; ++I)
                    acc[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[I >> 1][I & 1], b, acc[I], 0, 0, 0);
                // two requests of the next stage's image in the shadow of this k-slice's MFMAs
                if (Hnext) {
                    if (2 * kt < NDMA) request(Hnext, nbuf, 2 * kt);
                    if (2 * kt + 1 < NDMA) request(Hnext, nbuf, 2 * kt + 1);
                }
                // last stage: row 4 kt + rg (+ 2 D^2: the p block) of the next item's tile
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 3; ++i) a_cur[i] = a_nxt[i];
            }
            MONO_TICK(3);                                                    // product loop (issue)
            const double wgt = (st == 0 || st == 3) ? 1.0 : 2.0, c = (st == 2) ? dt : hh;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double kx = wmr[t][r] * Ys[t][r], ky = -acc[t][r];
                    SX[t][r] = fma(wgt, kx, SX[t][r]); SY[t][r] = fma(wgt, ky, SY[t][r]);
                    Xs[t][r] = fma(c, kx, X0[t][r]);
                    Ys[t][r] = fma(c, ky, Y0[t][r]);
                }
        }
        MONO_TICK(4);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (colok && 16 * t + rg + 4 * r < D) {
                    (Mx + (16 * t + 4 * r) * D)[toff] = fma(h6, SX[t][r], X0[t][r]);
                    (My + (16 * t + 4 * r) * D)[toff] = fma(h6, SY[t][r], Y0[t][r]);
                }
            }
        MONO_TICK(5);                                                        // stores of the tile's rows (issue)
        item = nitem;
#ifdef GDML_PHASE_CLOCK
        if (pc_on) pc[7] += 1;
#endif
    }
#ifdef GDML_PHASE_CLOCK
    if (pc_on) for (int i = 0; i < 8; ++i) g_mono_clock[i] = pc[i];
#endif
}

template <int NT, int KT>
__global__ __launch_bounds__(256, 1) void dense_mono_mfma_slab_dma(MonoArgs A) { dense_mono_mfma_slab_body<NT, KT, true>(A); }
template <int NT, int KT>
__global__ __launch_bounds__(256, 1) void dense_mono_mfma_slab_reg(MonoArgs A) { dense_mono_mfma_slab_body<NT, KT, false>(A); }

// prefactor matrix, determinant and branch tracker from the monodromy blocks in global memory
__global__ __launch_bounds__(256) void dense_prefactor_kernel(MonoArgs A) {
    extern __shared__ double2 smem2[];
    __shared__ int ipiv;
    const int D = A.st.dim, DD = D * D, tid = threadIdx.x, nth = blockDim.x, dp = A.hk.dprime;
    cplx *mat = (cplx *)smem2, *X = mat + (size_t)dp * dp;
    if (A.fixup && A.st.flags[A.st.n] == 0) return;          // nothing was flagged in this step
    for (int64_t tr = blockIdx.x; tr < A.st.n; tr += gridDim.x) {
        if (A.fixup && A.st.flags[tr] == 0) continue;        // uniform over the workgroup
        const double *M = A.st.mono + tr * 4 * (int64_t)DD;
        __syncthreads();
        if (A.hk.diag) {
            for (int e = tid; e < DD; e += nth) {
                const int a = e / D, b = e - a * D;
                const double sta = A.hk.st[a], sib = A.hk.si[b];
                mat[e] = c_make(0.5 * (sta / sib * M[e] + sib / sta * M[3 * DD + e]),
                                0.5 * (-SC_HBAR * sta * sib * M[DD + e] + M[2 * DD + e] / (SC_HBAR * sta * sib)));
            }
            __syncthreads();
        } else {
            general_prefactor_matrix(A.hk, M, M + DD, M + 2 * DD, M + 3 * DD, D, X, mat, A.panel);
        }
        const cplx det = lds_lu_det(mat, dp, &ipiv);
        if (tid == 0) {
            cplx *c2 = (cplx *)A.st.c2;
            if (A.mode == 0) {
                const cplx prev = c2[tr];
                if (prev.x < 0.0 && det.x < 0.0 && prev.y * det.y < 0.0) A.st.sgn[tr] = -A.st.sgn[tr];
            } else {
                A.st.sgn[tr] = 1.0;
            }
            c2[tr] = det;
            if (A.fixup) A.st.flags[tr] = 0;
        }
    }
}

// The same determinant with the matrix in REGISTERS for diagonal width matrices and D <= 96 (round 2): the block-pivoted
// dataflow elimination of the separable fast path (sc_hk_lu.h) with NR = ceil(D/16) <= 6 row / column slots per thread.
// The LDS kernel above needs 16 d'^2 bytes per trajectory (one workgroup per CU at D = 90) and two barriers per pivot:
// 10 ms at D = 90, n = 1e4.  Trajectories whose in-block pivot is too weak are flagged and redone by the LDS kernel.
template <int NR, int KB, int RW, class Barrier>
__device__ __forceinline__ void eliminate_all_blocks(cplx (&m)[NR][NR], cplx *detbuf, int D, int seq0, cplx (*rowbuf)[RW],
                                                     PivotRecord *pivrec, int *weak, int tid, Barrier &&barrier) {
    if constexpr (KB < NR) {
        eliminate_block<NR, KB, RW>(m, detbuf, D, seq0 + 1 + KB, rowbuf, pivrec, weak, tid, barrier);
        eliminate_all_blocks<NR, KB + 1, RW>(m, detbuf, D, seq0, rowbuf, pivrec, weak, tid, barrier);
    }
}

template <int NR>
__global__ __launch_bounds__(256, NR > 4 ? 2 : 4) void dense_prefactor_reg_kernel(MonoArgs A) {
    constexpr int RW = 16 * NR;
    __shared__ cplx rowbuf[16][RW];
    __shared__ PivotRecord pivrec[16];
    __shared__ cplx detbuf[16 * NR];
    __shared__ int weak;                     // bit 0: weak in-block pivot, bit 1: zero pivot
    __shared__ double scl[4 * RW];           // st, 1/st, si, 1/si
    const int D = A.st.dim, DD = D * D, tid = threadIdx.x, tj = tid & 15;
    const int trow = (tid >> 6) * 4 + ((tid >> 4) & 3);
    if (tid < 16) pivrec[tid].pad = 0;
    if (tid < RW) {
        const double st = tid < D ? A.hk.st[tid] : 1.0, si = tid < D ? A.hk.si[tid] : 1.0;
        scl[tid] = st; scl[RW + tid] = 1.0 / st; scl[2 * RW + tid] = si; scl[3 * RW + tid] = 1.0 / si;
    }
    __syncthreads();
    auto barrier = [] { __syncthreads(); };
    const bool rows_odd = row_order_is_odd(D);
    const unsigned to = (unsigned)(trow * D + tj);
    int seq0 = 0;
    for (int64_t tr = blockIdx.x; tr < A.st.n; tr += gridDim.x, seq0 += NR) {
        const double *M = A.st.mono + tr * 4 * (int64_t)DD;
        if (tid == 0) weak = 0;
        if (tid < 16 * NR) detbuf[tid] = c_make(1.0, 0.0);
        int til = trow, tjl = tj;
        __asm__ volatile("" : "+v"(til), "+v"(tjl));            // LDS indices recomputed per trajectory, not hoisted and spilled
        cplx m[NR][NR];
#pragma unroll
        for (int ra = 0; ra < NR; ++ra) {
            const int a = 16 * ra + til;
            const double sa = scl[a], isa = scl[RW + a];
            double v[4][NR];
#pragma unroll
            for (int rb = 0; rb < NR; ++rb) {
                const bool ok = a < D && 16 * rb + tj < D;
                const double *pe = M + __builtin_amdgcn_readfirstlane(16 * ra * D + 16 * rb);     // wave-uniform base + one 32-bit offset
                v[0][rb] = ok ? pe[to] : 0.0; v[1][rb] = ok ? pe[DD + to] : 0.0;
                v[2][rb] = ok ? pe[2 * DD + to] : 0.0; v[3][rb] = ok ? pe[3 * DD + to] : 0.0;
            }
#pragma unroll
            for (int rb = 0; rb < NR; ++rb) {
                const int b = 16 * rb + tjl;
                const double sb = scl[2 * RW + b], isb = scl[3 * RW + b];
                m[ra][rb] = c_make(0.5 * (sa * isb * v[0][rb] + isa * sb * v[3][rb]),
                                   0.5 * (-SC_HBAR * sa * sb * v[1][rb] + (1.0 / SC_HBAR) * isa * isb * v[2][rb]));
                // the element must exist before the next slot's loads are issued: otherwise every raw value of the
                // trajectory (4 NR^2 doubles) is loaded first and spilled
                __asm__ volatile("" : "+v"(m[ra][rb].x), "+v"(m[ra][rb].y) : : "memory");
            }
        }
        eliminate_all_blocks<NR, 0, RW>(m, detbuf, D, seq0, rowbuf, pivrec, &weak, tid, barrier);
        __syncthreads();
        if (tid < 16) lu_partial_products<NR>(detbuf, tid);
        if (tid == 0) {
            if (weak) {
                A.st.flags[tr] = 1;                  // c2 / sgn are left to the fully pivoted LDS elimination
                atomicAdd(&A.st.flags[A.st.n], 1);
            } else {
                const cplx det = finish_determinant(detbuf, rows_odd);
                cplx *c2 = (cplx *)A.st.c2;
                if (A.mode == 0) {
                    const cplx prev = c2[tr];
                    if (prev.x < 0.0 && det.x < 0.0 && prev.y * det.y < 0.0) A.st.sgn[tr] = -A.st.sgn[tr];
                } else {
                    A.st.sgn[tr] = 1.0;
                }
                c2[tr] = det;
            }
        }
        __syncthreads();
    }
}

// ---- RK4 of (q, p, S) for potentials the CALLER evaluates (generic Python potentials: SURVEY.md section 8b, "generic
// Python potentials take the unfused path").  Per stage: sc_stage_point writes the stage positions, the caller
// evaluates V, grad, hess there with its own code, sc_stage_consume does the bookkeeping of the reference's RK4
// (propagators.py:86-119) and of EquationsOfMotion.f (:313-383: q' = p/m, p' = -grad, S' = T - V, <T+V> at the
// k4 stage).  One wavefront per trajectory, lane = coordinate.
struct StageArgs2 {
    sc_state st;
    sc_dense_scratch sc;
    const double *inv_mass, *V, *grad;      // V [n], grad [n][D] (trajectory-major)
    double *r_out;                           // [n][D]
    double dt;
    int stage;
    double *epart;
};

__global__ __launch_bounds__(256) void stage_point_kernel(StageArgs2 A) {
    const int D = A.st.dim, s = A.stage;
    const double c = (s == 0) ? 0.0 : (s == 3 ? A.dt : 0.5 * A.dt);
    const int64_t total = A.st.n * D;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t tr = e / D;
        const int i = (int)(e - tr * D);
        A.r_out[e] = A.st.qp[tr * 2 * D + i] + (s ? c * A.sc.kprev[tr * 2 * D + i] : 0.0);
    }
}

__global__ __launch_bounds__(256) void stage_consume_kernel(StageArgs2 A) {
    const int D = A.st.dim, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = A.stage;
    const double dt = A.dt, c = (s == 0) ? 0.0 : (s == 3 ? dt : 0.5 * dt), w = (s == 0 || s == 3) ? 1.0 : 2.0;
    const double h6 = dt / 6.0;
    __shared__ double wsum[4];
    double esum = 0.0;
    for (int64_t tr = (int64_t)blockIdx.x * 4 + wave; tr < A.st.n; tr += (int64_t)gridDim.x * 4) {
        double *qp = A.st.qp + tr * 2 * D;
        double *kprev = A.sc.kprev + tr * 2 * D, *ksum = A.sc.ksum + tr * 2 * D;
        const double *g = A.grad + tr * D;
        double tk = 0.0;
        for (int i = lane; i < D; i += 64) {
            const double ps = qp[D + i] + (s ? c * kprev[D + i] : 0.0);       // momentum at the stage point
            const double im = A.inv_mass[i], kq = ps * im, kp = -g[i];
            tk += 0.5 * ps * ps * im;
            kprev[i] = kq; kprev[D + i] = kp;
            const double sq = (s ? ksum[i] : 0.0) + w * kq, sp = (s ? ksum[D + i] : 0.0) + w * kp;
            if (s < 3) { ksum[i] = sq; ksum[D + i] = sp; }
            else { qp[i] += h6 * sq; qp[D + i] += h6 * sp; }
        }
        tk = wave_sum(tk);
        if (lane == 0) {
            const double e = A.V[tr], ds = tk - e, acc = (s ? A.sc.ssum[tr] : 0.0) + w * ds;
            if (s < 3) A.sc.ssum[tr] = acc;
            else { A.st.act[tr] += h6 * acc; esum += tk + e; }
        }
    }
    if (lane == 0) wsum[wave] = esum;
    __syncthreads();
    if (threadIdx.x == 0 && A.epart && s == 3) A.epart[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

}  // namespace

#ifdef GDML_PHASE_CLOCK
extern "C" int sc_mono_phase_clock(unsigned long long *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_mono_clock), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#endif

extern "C" int sc_stage_point(const sc_state *st, const sc_dense_scratch *sc, double dt, int32_t stage, double *r_out,
                              void *stream) {
    if (!st || !sc || !sc->kprev || !r_out) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_stage_point: null argument");
    if (stage < 0 || stage > 3) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_stage_point: stage %d", stage);
    if (st->n <= 0) return SC_OK;
    StageArgs2 a{*st, *sc, nullptr, nullptr, nullptr, r_out, dt, stage, nullptr};
    const int64_t blocks = (st->n * st->dim + 255) / 256;
    hipLaunchKernelGGL(stage_point_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, a);
    return sc_check_launch("sc_stage_point");
}

extern "C" int sc_stage_consume(const sc_state *st, const sc_dense_scratch *sc, const double *inv_mass, const double *V,
                                const double *grad, double dt, int32_t stage, double *energy_partials, void *stream) {
    if (!st || !sc || !sc->kprev || !sc->ksum || !sc->ssum || !inv_mass || !V || !grad)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_stage_consume: null argument");
    if (stage < 0 || stage > 3) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_stage_consume: stage %d", stage);
    if (st->n <= 0) return SC_OK;
    StageArgs2 a{*st, *sc, inv_mass, V, grad, nullptr, dt, stage, energy_partials};
    hipLaunchKernelGGL(stage_consume_kernel, dim3(sc_dense_grid(st->n)), dim3(256), 0, (hipStream_t)stream, a);
    return sc_check_launch("sc_stage_consume");
}

int sc_launch_dense_any(const sc_state *st, const sc_hk_consts *hk, const double *inv_mass, const double *hess,
                        double *scratch, double dt, int mode, hipStream_t s);        // sc_dense_any.hip

extern "C" int sc_dense_mono_step(const sc_state *st, const sc_hk_consts *hk, const double *inv_mass, const double *hess,
                                  double *mono_sums, double dt, int32_t mode, void *stream) {
    if (!st || !hk || (mode == 0 && (!inv_mass || !hess)))
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_dense_mono_step: null argument");
    const int D = st->dim;
    if (int rq = sc_require_rowmajor(st, "sc_dense_mono_step")) return rq;
    if (D < 1) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_dense_mono_step: D=%d", D);
    if (D > 64 && (mode == 0 || D > 96) && !mono_sums)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_dense_mono_step: D=%d > 64 needs the mono_sums scratch "
                       "(sc_dense_mono_scratch_bytes)", D);
    if (hk->dim != D || hk->dprime < 1 || hk->dprime > D)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_dense_mono_step: bad prefactor constants");
    if (st->n <= 0) return SC_OK;
    hipStream_t s = (hipStream_t)stream;
    if (D > 96) return sc_launch_dense_any(st, hk, inv_mass, hess, mono_sums, dt, mode, s);    // no size limit, slow
    // prefactor kernel: d' x d' matrix + a panel of X = M R in LDS
    const size_t dp = hk->dprime, budget = 150 * 1024;
    int panel = (int)dp;
    if (!hk->diag) while (panel > 8 && (dp * dp + (size_t)D * panel) * 16 + 32 > budget) panel = (panel + 1) / 2;
    const size_t lds = (dp * dp + (hk->diag ? 0 : (size_t)D * panel)) * 16 + 32;
    if (lds > 160 * 1024) return sc_fail(SC_ERR_UNSUPPORTED, "sc_dense_mono_step: needs %zu B of LDS", lds);
    MonoArgs a{*st, *hk, inv_mass, hess, 4 * (int64_t)D * D, (int64_t)D * D, mono_sums, panel, dt, mode, 0};
    if (mode == 0) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 256;
        const int nt = (D + 15) / 16;
        bool use_big = false;                     // the global-scratch kernel: kept for A/B in the tuning build
#ifdef SC_TUNING
        use_big = nt > 4 && getenv("SC_MONO_BIG") != nullptr;
#endif
        const size_t rk4_lds = nt <= 4 ? ((size_t)2 * 16 * nt * HS + 64 + (size_t)2 * nt * (nt < 4 ? 2 : 1) * 4 * nt * 64) * sizeof(double)
                                       : ((size_t)16 * nt * HSB + 128) * sizeof(double);
#define SC_LAUNCH_MONO(KERNEL_, NT_, KT_, WGS_, THREADS_)                                                          \
        do {                                                                                                        \
            if (hipFuncSetAttribute((const void *)KERNEL_<NT_, KT_>,                                                \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)rk4_lds) != hipSuccess)        \
                return sc_check_launch("sc_dense_mono_step (LDS attribute)");                                       \
            const int64_t g_ = (int64_t)cus * (WGS_), items_ = st->n * (nt > 4 && !use_big ? (2 * nt + 3) / 4 : 1); \
            hipLaunchKernelGGL((KERNEL_<NT_, KT_>), dim3((unsigned)(items_ < g_ ? items_ : g_)),                    \
                               dim3(THREADS_), rk4_lds, s, a);                                                      \
        } while (0)
        // slab kernel: Hessian images by LDS-DMA when the rows are 16-byte aligned (D even, 16-byte aligned base and strides)
        const bool dma_ok = (D % 2 == 0) && (((uintptr_t)hess & 15) == 0);
        const size_t lds2 = (size_t)(2 * D + 2) * HSB * sizeof(double);      // two packed images (dense_mono_mfma_slab_dma2)
        const bool two_images = lds2 <= 160 * 1024;
#define SC_LAUNCH_SLAB(NT_, KT_)                                                                                    \
        do {                                                                                                        \
            if (dma_ok && two_images) {                                                                             \
                const size_t keep_ = rk4_lds;                                                                       \
                (void)keep_;                                                                                        \
                if (hipFuncSetAttribute((const void *)dense_mono_mfma_slab_dma2<NT_, KT_>,                          \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)       \
                    return sc_check_launch("sc_dense_mono_step (LDS attribute)");                                   \
                const int64_t g_ = (int64_t)cus * 4, items_ = st->n * ((2 * nt + 3) / 4);                           \
                hipLaunchKernelGGL((dense_mono_mfma_slab_dma2<NT_, KT_>), dim3((unsigned)(items_ < g_ ? items_ : g_)), \
                                   dim3(256), lds2, s, a);                                                          \
            } else if (dma_ok) SC_LAUNCH_MONO(dense_mono_mfma_slab_dma, NT_, KT_, 4, 256);                          \
            else SC_LAUNCH_MONO(dense_mono_mfma_slab_reg, NT_, KT_, 4, 256);                                        \
        } while (0)
#define SC_SLAB_CASES(NT_)                                                                                          \
        case 4 * NT_ - 3: SC_LAUNCH_SLAB(NT_, 4 * NT_ - 3); break;                                                  \
        case 4 * NT_ - 2: SC_LAUNCH_SLAB(NT_, 4 * NT_ - 2); break;                                                  \
        case 4 * NT_ - 1: SC_LAUNCH_SLAB(NT_, 4 * NT_ - 1); break;                                                  \
        case 4 * NT_: SC_LAUNCH_SLAB(NT_, 4 * NT_); break;
#define SC_MONO_CASES(KERNEL_, NT_, WGS_, THREADS_)                                                                 \
        case 4 * NT_ - 3: SC_LAUNCH_MONO(KERNEL_, NT_, 4 * NT_ - 3, WGS_, THREADS_); break;                         \
        case 4 * NT_ - 2: SC_LAUNCH_MONO(KERNEL_, NT_, 4 * NT_ - 2, WGS_, THREADS_); break;                         \
        case 4 * NT_ - 1: SC_LAUNCH_MONO(KERNEL_, NT_, 4 * NT_ - 1, WGS_, THREADS_); break;                         \
        case 4 * NT_: SC_LAUNCH_MONO(KERNEL_, NT_, 4 * NT_, WGS_, THREADS_); break;
        if (use_big) {
            switch ((D + 3) / 4) {
                SC_MONO_CASES(dense_mono_mfma_big_kernel, 5, 1, 320)
                SC_MONO_CASES(dense_mono_mfma_big_kernel, 6, 1, 384)
            }
        } else {
            switch ((D + 3) / 4) {
                SC_MONO_CASES(dense_mono_mfma_kernel, 1, 8, 128)
                SC_MONO_CASES(dense_mono_mfma_kernel, 2, 4, 256)
                SC_MONO_CASES(dense_mono_mfma_kernel, 3, 1, 384)
                SC_MONO_CASES(dense_mono_mfma_kernel, 4, 1, 512)
                SC_SLAB_CASES(5)
                SC_SLAB_CASES(6)
            }
        }
#undef SC_MONO_CASES
#undef SC_SLAB_CASES
#undef SC_LAUNCH_SLAB
#undef SC_LAUNCH_MONO
        const int rc = sc_check_launch("sc_dense_mono_step (RK4)");
        if (rc) return rc;
    }
    bool reg_lu = hk->diag && st->flags != nullptr;       // diagonal widths: determinant in registers, LDS kernel as the fix-up
#ifdef SC_TUNING
    if (getenv("SC_PREFACTOR_LDS")) reg_lu = false;
#endif
    if (reg_lu) {
        if (hipMemsetAsync(st->flags + st->n, 0, sizeof(int32_t), s) != hipSuccess)
            return sc_check_launch("sc_dense_mono_step (flag counter)");
        const int64_t cap = 1024;
        const unsigned grid = (unsigned)(st->n < cap ? st->n : cap);
        switch ((D + 15) / 16) {
            case 1: hipLaunchKernelGGL(dense_prefactor_reg_kernel<1>, dim3(grid), dim3(256), 0, s, a); break;
            case 2: hipLaunchKernelGGL(dense_prefactor_reg_kernel<2>, dim3(grid), dim3(256), 0, s, a); break;
            case 3: hipLaunchKernelGGL(dense_prefactor_reg_kernel<3>, dim3(grid), dim3(256), 0, s, a); break;
            case 4: hipLaunchKernelGGL(dense_prefactor_reg_kernel<4>, dim3(grid), dim3(256), 0, s, a); break;
            case 5: hipLaunchKernelGGL(dense_prefactor_reg_kernel<5>, dim3(grid), dim3(256), 0, s, a); break;
            default: hipLaunchKernelGGL(dense_prefactor_reg_kernel<6>, dim3(grid), dim3(256), 0, s, a); break;
        }
        const int rc = sc_check_launch("sc_dense_mono_step (register prefactor)");
        if (rc) return rc;
        a.fixup = 1;
    }
    if (hipFuncSetAttribute((const void *)dense_prefactor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return sc_check_launch("sc_dense_mono_step (LDS attribute)");
    hipLaunchKernelGGL(dense_prefactor_kernel, dim3(sc_dense_grid(st->n)), dim3(256), lds, s, a);
    return sc_check_launch("sc_dense_mono_step");
}
