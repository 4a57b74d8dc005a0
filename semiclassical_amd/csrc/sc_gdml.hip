// sGDML force field (energy, gradient, analytic Hessian) and the RK4 step for potentials with a dense,
// position-dependent Hessian.
//
// Reference semantics reproduced (paths relative to the reference repository):
//   GDMLPredict.forward                     semiclassical/gdml_predictor.py:96-250
//   MolecularGDMLPotential                  semiclassical/potentials.py:641-744
//   RK4 + equations of motion               semiclassical/propagators.py:86-119, 313-383
//   HK prefactor + branch tracking          semiclassical/propagators.py:951-1052
//
// Structure of one time step (the Hessian depends on the stage position, so the four stages cannot be fused):
//   sc_gdml_stage(s), s = 0..3   one workgroup per trajectory: stage point (q_s, p_s) from the previous slope,
//                                V, grad V, hess V of the sGDML model there; slopes of (q, p, S); hess V -> scratch;
//                                after stage 3 the new (q, p, S) and the <T+V> partial sums
//   sc_dense_mono_step           one workgroup per trajectory: RK4 of the four monodromy blocks with the four stage
//                                Hessians (D x D times D x 2D products, Q stage matrix and Hessian staged in LDS,
//                                the thread's elements and RK4 accumulators in registers), prefactor matrix,
//                                pivoted LU determinant, branch tracking
//
// sGDML evaluation per geometry (Dd = N(N-1)/2 inverse distances x_d, M training points, q = sqrt(5)/sigma):
//   xd_m = x - x_m,  d_m = |xd_m|,  XA_m = xd_m . A_m,  e_m = q^4/3 exp(-q d_m),  f_m = e_m (1 + q d_m)/q^2
//   E = std sum_m f_m XA_m + c
//   g_x = sum_m f_m A_m - e_m XA_m xd_m                        grad = std J^T g_x
//   hess = std [ sum_m w_m XJ_m XJ_m^T - e_m (AJ_m XJ_m^T + XJ_m AJ_m^T) - (sum_m e_m XA_m) J^T J + sum_d g_x[d] d2x_d ]
//   with w_m = e_m XA_m q / d_m, XJ_m = J^T xd_m, AJ_m = J^T A_m.  The Jacobian row of pair d = (k, l) is
//   jd = -x_d^3 (r_k - r_l) on atom k and -jd on atom l, so J^T v is a gather over the N-1 partners of an atom,
//   J^T J and the second-derivative term are 3x3 blocks per atom pair.
// The three rank-M sums are ONE GEMM on the FP64 matrix cores (v_mfma_f64_16x16x4_f64): with Z_m = w_m XJ_m - e_m AJ_m
//   sum_m w_m XJ_m XJ_m^T - e_m (AJ_m XJ_m^T + XJ_m AJ_m^T) = [XJ ; -e AJ]^T [Z ; XJ]        (3N x 2M times 2M x 3N)
// accumulated per chunk of training points from LDS operands, one 16 x 16 tile of the upper triangle per accumulator
// (tiles dealt round-robin to the wavefronts); the atom-pair terms are added to the accumulators element by element.
#include "sc_common.h"
#include "sc_prefactor.h"
#include "sc_row16.h"

namespace {

__device__ __forceinline__ int pair_index(int a, int b) {      // a != b; torch.tril_indices order (i > j)
    return a > b ? a * (a - 1) / 2 + b : b * (b - 1) / 2 + a;
}

struct GdmlLds {
    double *pos, *x, *r, *gx, *fm, *em, *wm, *ea, *grad, *P, *Qn, *Z, *dg, *stage, *red;
    int XP;      // row stride of the MFMA operand arrays P = XJ, Qn = -e AJ, Z = w XJ - e AJ  ([GDML_CH][XP])
};

// Launch shapes (threads, training points per chunk, accumulator tiles per wavefront, stage buffers) by molecule size:
//   3N <= 64 (10 Hessian tiles)    four wavefronts, CH = 4, three tiles per wavefront; two or three geometries share a CU
//   22 .. 32 atoms                 eight wavefronts, three tiles per wavefront, CH = 8 (one training point per wavefront in
//                                  the row reductions) while the stage fits twice, else CH = 4
//   33 .. 40 atoms (36 tiles)      eight wavefronts, five tiles, CH = 4
//   41 .. 48 atoms (45 tiles)      eight wavefronts, six tiles, CH = 4, ONE stage buffer (the copy of chunk k + 1 is started
//                                  behind the last reader of chunk k and runs under the matrix-core phase only)
// (round 3: the reference's predictor has no size limit, gdml_predictor.py:96-250).  Measured and rejected at 30 atoms: four
// wavefronts with six tiles each and CH = 4, so that two geometries share a CU and fill each other's barrier waits (77 KB of
// LDS each): 6.12 ms per stage launch against 5.51 ms of the eight-wavefront shape (half the training points per chunk
// amortise the same per-chunk work, 59 spilled registers).  Every N of a row-length bucket
// (sc_gdml_row_len) resolves to an instantiated kernel: the dispatch below matches (threads, row length, CH, NB).
struct GdmlShape { int threads, ch, mt, nb; };
size_t gdml_lds_doubles_ch(int N, int Dd, int ch, int nb);
extern "C" int sc_gdml_row_len(int32_t n_atoms);
inline GdmlShape gdml_shape(int N) {
    const int Dd = N * (N - 1) / 2;
    const size_t cu = 160 * 1024;
    if (3 * N <= 64) return {256, 4, 3, 2};
    if (N <= 32) {
        return {512, gdml_lds_doubles_ch(N, Dd, 8, 2) * 8 <= cu ? 8 : 4, 3, 2};
    }
    if (N <= 40) return {512, 4, 5, 2};
    return {512, 4, 6, 1};
}

// row stride of the operand arrays: 16 T (+16) doubles with stride = 16 mod 32, so that the four rows an MFMA operand
// read touches (64 lanes x 8 bytes) fall into different LDS banks
__host__ __device__ inline int gdml_xp(int N) {
    const int T = (3 * N + 15) / 16;
    return (T & 1) ? 16 * T : 16 * T + 16;
}

__device__ GdmlLds gdml_carve(double *base, int N, int Dd, int GDML_CH) {
    GdmlLds L;
    double *f = base;
    L.red = f;  f += 32;
    L.pos = f;  f += 3 * N;
    L.x = f;    f += Dd;
    L.r = f;    f += Dd;           // distances 1 / x
    L.gx = f;   f += Dd;
    L.fm = f;   f += GDML_CH;
    L.em = f;   f += GDML_CH;
    L.wm = f;   f += GDML_CH;
    L.ea = f;   f += GDML_CH;
    L.grad = f; f += 3 * N + (N & 1);
    L.dg = f;   f += 9 * N + (N & 1);
    L.XP = gdml_xp(N);
    L.P = f;    f += GDML_CH * L.XP;
    L.Qn = f;   f += GDML_CH * L.XP;
    L.Z = f;    f += GDML_CH * L.XP;
    if ((f - base) & 1) ++f;           // 16-byte aligned: the stage is filled by 16-byte LDS-DMA loads
    L.stage = f;                       // [NB][2][GDML_CH][Dd]: buffers of a chunk's rows of xs_train, then of jx_alphas
    return L;
}

size_t gdml_lds_doubles_ch(int N, int Dd, int GDML_CH, int nb) {
    return 32 + 3 * N + 3 * (size_t)Dd + 4 * GDML_CH + 3 * N + (N & 1) + 9 * N + (N & 1) + 3 * (size_t)GDML_CH * gdml_xp(N) +
           (size_t)nb * 2 * (size_t)GDML_CH * Dd + 1;       // + 1: alignment pad of the stage
}
size_t gdml_lds_doubles(int N, int Dd) { const GdmlShape sh = gdml_shape(N); return gdml_lds_doubles_ch(N, Dd, sh.ch, sh.nb); }

// V (without origin), grad[3N] (LDS, L.grad) and hess[3N][3N] (global, row-major) at the geometry in L.pos.
// Every thread returns the energy.  256 threads.
//
// ONE pass over the training set: a chunk of GDML_CH training points (their rows of xs_train and jx_alphas are one
// contiguous span each) is staged in LDS -- requested one chunk ahead into registers -- and serves everything that needs
// it: the row reductions d_m, XA_m (one wavefront per training point), the descriptor-space gradient (a thread per
// descriptor element, Neumaier-compensated over ALL training points in order), and the operands XJ_m, AJ_m of the Hessian
// GEMM (gathered from the staged rows with the thread's Jacobian coefficients in registers).  Round 2 read the training
// set three times per geometry (5.9 MB from L2 at 30 atoms / 200 points, the kernel's bound); this reads 1.4 MB.
// HN = half of the partner atoms the instantiation holds (four threads share the partners of one atom), 2 HN >= N.
// MT = accumulator tiles per wavefront, NB = stage buffers (see gdml_tiles_per_wave, gdml_nb)
// -DGDML_PHASE_CLOCK (variant library, tools/gdml_phases.py): wave 0 of workgroup 0 accumulates the shader cycles it spends in
// every phase of its geometries into phase_clock[0..9]
#ifdef GDML_PHASE_CLOCK
__device__ unsigned long long *g_phase_clock = nullptr;
#define GDML_TICK(slot) do { if (pc_on) { const unsigned long long now_ = clock64(); pc[slot] += now_ - pc_last; pc_last = now_; } } while (0)
#else
#define GDML_TICK(slot) do { } while (0)
#endif

template <int HN, int THREADS, int GDML_CH, int MT, int NB>
__device__ double gdml_eval_device(const sc_gdml_model &G, const GdmlLds &L, double *hess) {
#ifdef GDML_PHASE_CLOCK
    const bool pc_on = blockIdx.x == 0 && threadIdx.x == 0 && g_phase_clock != nullptr;
    unsigned long long pc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pc_last = clock64();
#endif
    constexpr int GDML_MAX_TILES = MT;
    constexpr int nth = THREADS, nw = THREADS / 64;
    const int N = G.n_atoms, Dd = G.n_desc, Mt = G.n_train, X = 3 * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double q = G.q;
    // The chunk's rows (one contiguous span of GDML_CH Dd doubles per array) go from L2 straight into the stage with
    // 16-byte LDS-DMA loads (global_load_lds_dwordx4: no registers in between, a wave instruction fills 1 KB); a partial
    // last chunk is copied through registers and padded with zeros.  The copy of chunk k + 1 is started behind the last
    // reader of chunk k and runs under the matrix-core phase.
    auto stage_chunk = [&](int m0, int buf) {
        double *stage = L.stage + buf * 2 * GDML_CH * Dd;
        const int mc = min(GDML_CH, Mt - m0);
        if (mc == GDML_CH) {
            const int units = GDML_CH * Dd / 2;                      // 16-byte units per array (GDML_CH Dd is even)
            for (int arr = 0; arr < 2; ++arr) {
                const double *src = (arr ? G.jx_alphas : G.xs_train) + (size_t)m0 * Dd;
                double *dst = stage + arr * GDML_CH * Dd;
                for (int u0 = 64 * wave; u0 < units; u0 += 64 * nw) {
                    if (u0 + lane < units)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 2 * (u0 + lane)),
                                                         (__attribute__((address_space(3))) void *)(dst + 2 * u0), 16, 0, 0);
                }
            }
        } else {
            for (int i = tid; i < 2 * GDML_CH * Dd; i += nth) {
                const int arr = i >= GDML_CH * Dd, ii = arr ? i - GDML_CH * Dd : i;
                stage[i] = ii < mc * Dd ? (arr ? G.jx_alphas : G.xs_train)[(size_t)m0 * Dd + ii] : 0.0;
            }
        }
    };
    stage_chunk(0, 0);
    // ---- descriptor and Jacobian rows
    for (int d = tid; d < Dd; d += nth) {
        const int k = G.pair_k[d], l = G.pair_l[d];
        const double dx = L.pos[3 * k] - L.pos[3 * l], dy = L.pos[3 * k + 1] - L.pos[3 * l + 1],
                     dz = L.pos[3 * k + 2] - L.pos[3 * l + 2];
        const double dist = sqrt(dx * dx + dy * dy + dz * dz);
        L.r[d] = dist;
        L.x[d] = 1.0 / dist;
    }
    for (int e = tid; e < 3 * GDML_CH * L.XP; e += nth) L.P[e] = 0.0;       // P, Qn, Z are contiguous: padding columns stay 0
    __syncthreads();
    // Jacobian coefficient of pair (a, c) on atom a, component u: d x_(a,c) / d r_a[u] = -x^3 (r_a - r_c)[u] (computed where it
    // is needed -- once per geometry -- instead of kept in LDS: 3 Dd doubles less per workgroup)
    auto jac = [&](double x3, int a, int c, int u) { return -x3 * (L.pos[3 * a + u] - L.pos[3 * c + u]); };
    // accumulator tiles of the rank-M sums on the matrix cores.  Tile t = c (c + 1) / 2 + r, r <= c, of the upper triangle
    // belongs to wavefront t % nw; accumulator layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 reg.
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int XP = L.XP, T = (X + 15) / 16, ntiles = T * (T + 1) / 2, rg = lane >> 4, li = lane & 15;
    d4 acc[GDML_MAX_TILES];
    int tr_[GDML_MAX_TILES], tc_[GDML_MAX_TILES];
#pragma unroll
    for (int sl = 0; sl < GDML_MAX_TILES; ++sl) {
        acc[sl] = (d4){0.0, 0.0, 0.0, 0.0};
        // a slot beyond the last tile repeats the last tile (its accumulator is never written out): the matrix-core
        // phase below has no branches
        const int t = min(__builtin_amdgcn_readfirstlane(wave) + sl * nw, ntiles - 1);
        int c = 0;
        while ((c + 1) * (c + 2) / 2 <= t) ++c;
        tr_[sl] = t - c * (c + 1) / 2; tc_[sl] = c;
    }
    // formation roles: thread = (MPT of the chunk's training points, atom, quarter of the partner atoms); its
    // Jacobian coefficients coef[u][c] = +-jd[pair(a, c)][u] stay in registers (3 QN doubles):
    //   XJ_m[a, u] = base[u] - sum_c coef[u][c] xs_m[pair(a, c)],   AJ_m[a, u] = sum_c coef[u][c] A_m[pair(a, c)]
    // the four partial sums of a row sit in adjacent lanes and meet by two quad-permute additions
    constexpr int PARTS = 4, QN = (2 * HN + PARTS - 1) / PARTS;
    constexpr int MPT = GDML_CH * 4 * 32 > THREADS * 2 ? 4 : 2;     // training points per thread: 4 N (GDML_CH / MPT) <= THREADS
    const int fm_q = tid % PARTS, fm_at = (tid / PARTS) % N, fm_mh = (tid / PARTS) / N;
    const bool fm_active = fm_mh < GDML_CH / MPT;
    auto parts_sum = [](double v) {
        v += dpp_mov_f64<0xB1>(v);                       // quad_perm [1,0,3,2]
        v += dpp_mov_f64<0x4E>(v);                       // quad_perm [2,3,0,1]
        return v;
    };
    double coef[3][QN], base[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int cc = 0; cc < QN; ++cc) {
        const int c = QN * fm_q + cc;
        const bool ok = c < N && c != fm_at;
        const int d = ok ? pair_index(fm_at, c) : 0;
        const double xq = ok ? L.x[d] : 0.0, x3 = xq * xq * xq;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const double j = ok ? jac(x3, fm_at, c, u) : 0.0;
            coef[u][cc] = j;
            base[u] = fma(j, xq, base[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) base[u] = parts_sum(base[u]);     // complete in the lane of part 0
    // descriptor-space gradient g_x[d] = sum_m f_m A_m[d] - e_m XA_m (x[d] - xs_m[d]) of this thread's elements d = tid,
    // tid + 256.  The terms (2e8) cancel to the size of the force (6e1): Neumaier-compensated accumulation keeps the
    // rounding of the SUM out of the result (what remains is the rounding of the terms themselves)
    constexpr int EPT = (HN * (2 * HN - 1) + nth - 1) / nth;   // descriptor elements per thread (Dd <= HN (2 HN - 1) for N <= 2 HN atoms)
    double gacc[EPT], gcomp[EPT], xown[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) { gacc[j] = 0.0; gcomp[j] = 0.0; xown[j] = tid + nth * j < Dd ? L.x[tid + nth * j] : 0.0; }
    double esum = 0.0, ssum = 0.0;
    double sx[MPT][3], sa3[MPT][3];
    // (1) row reductions of the training points this wavefront owns in the chunk at m0: d_m^2 = |x - xs_m|^2,
    //     XA_m = (x - xs_m) . A_m (lanes over the descriptor), then the scalars e_m, f_m, w_m (gdml_predictor.py:150-170)
    const double iq2 = 1.0 / (q * q);
    // Branch-free: a wavefront without a training point in this chunk reduces the last row again and drops the result, and
    // every lane evaluates the scalar tail (the sums are wave-uniform), so that the gathers of (2) -- issued between the
    // reductions and the tail -- fill the latency of its dependent chain (rsq, Newton steps, exp).
    auto row_scalars = [&](int m0, const double *sxs, const double *sal, auto &&between) {
        const int mm = min(wave, GDML_CH - 1);
        const bool has = wave < min(GDML_CH, Mt - m0);
        double s2 = 0.0, sa = 0.0;
        {
            // two elements per trip (the six LDS reads of a trip are requested together; a trip used to wait for three reads per
            // element, one element after the other), two accumulator pairs
            double s2b = 0.0, sab = 0.0;
#ifdef GDML_ABLATE_ROWRED
            for (int d = lane; d < 64; d += 128) {
#else
            for (int d = lane; d < Dd; d += 128) {
#endif
                const int d2 = d + 64, dc = min(d2, Dd - 1);
                const double x0 = L.x[d], xs0 = sxs[mm * Dd + d], a0 = sal[mm * Dd + d];
                const double x1 = L.x[dc], xs1 = sxs[mm * Dd + dc], a1 = sal[mm * Dd + dc];
                const double xd0 = x0 - xs0, xd1 = d2 < Dd ? x1 - xs1 : 0.0;
                s2 = fma(xd0, xd0, s2); sa = fma(xd0, a0, sa);
                s2b = fma(xd1, xd1, s2b); sab = fma(xd1, a1, sab);
            }
            s2 += s2b; sa += sab;
        }
        s2 = wave_sum(s2); sa = wave_sum(sa);
        between();
#ifdef GDML_ABLATE_TAIL
        const double dist = 1.0 + s2, e = 1.0 - q * dist, rd = 1.0;
#else
        // 1/d from the hardware reciprocal square root + two Newton steps (full fp64 accuracy), d = s2 / d with one
        // correction (sqrt plus two fp64 divisions were two thirds of this serial tail)
        double rd = __builtin_amdgcn_rsq(s2);
        rd = fma(fma(-0.5 * s2 * rd, rd, 0.5), rd, rd);
        rd = fma(fma(-0.5 * s2 * rd, rd, 0.5), rd, rd);
        double dist = s2 * rd;
        dist = fma(fma(-dist, dist, s2), 0.5 * rd, dist);
        const double e = (1.0 / 3.0) * q * q * q * q * exp(-q * dist);
#endif
        const double f = e * (1.0 + q * dist) * iq2;
        if (has && lane == 0) {
            L.fm[mm] = f; L.em[mm] = e; L.wm[mm] = e * sa * q * rd; L.ea[mm] = e * sa;
            esum += f * sa; ssum += e * sa;
        }
    };
    // (2) J^T xs_m, J^T A_m of this thread's atom for its two training points (rows beyond a partial chunk are zero)
    const int fm_row = fm_active ? MPT * fm_mh : 0, tri = fm_at * (fm_at - 1) / 2;   // threads without a role gather rows 0.., unused
    int pidx[QN];
#pragma unroll
    for (int cc = 0; cc < QN; ++cc) {
        // position of pair (a, c) in a descriptor row; partners outside the molecule (and c = a) have coef = 0
        const int c = min(QN * fm_q + cc, N - 1);
#ifdef GDML_TEST_NOCONFLICT
        pidx[cc] = (tid & 31) + 32 * cc;      // experiment: conflict-free addresses (wrong results)
#else
        pidx[cc] = c < fm_at ? tri + c : (c > fm_at ? c * (c - 1) / 2 + fm_at : 0);
#endif
    }
    auto gather_begin = [&]() {
#pragma unroll
        for (int i = 0; i < MPT; ++i)
#pragma unroll
            for (int u = 0; u < 3; ++u) { sx[i][u] = 0.0; sa3[i][u] = 0.0; }
    };
    auto gather_partner = [&](auto ccc, const double *sxs, const double *sal) {
        constexpr int cc = decltype(ccc)::value;
#ifndef GDML_ABLATE_FORM
        const double *xr = sxs + fm_row * Dd, *ar = sal + fm_row * Dd;
#pragma unroll
        for (int i = 0; i < MPT; ++i) {
            const double xv = xr[i * Dd + pidx[cc]], av = ar[i * Dd + pidx[cc]];
#pragma unroll
            for (int u = 0; u < 3; ++u) { sx[i][u] = fma(coef[u][cc], xv, sx[i][u]); sa3[i][u] = fma(coef[u][cc], av, sa3[i][u]); }
        }
#endif
    };
    auto gather_end = [&]() {
#pragma unroll
        for (int i = 0; i < MPT; ++i)
#pragma unroll
            for (int u = 0; u < 3; ++u) { sx[i][u] = parts_sum(sx[i][u]); sa3[i][u] = parts_sum(sa3[i][u]); }
    };
    // Chunk loop, two stage buffers, two barriers per chunk:
    //     (1) row scalars, (2) gathers of chunk k | B1 | (3) gradient terms, (4) operand rows | B2 | (5) MFMAs
    // The copy of chunk k + 1 into the other buffer starts at the top of chunk k (the readers of that buffer finished
    // before B2 of chunk k - 1) and must have landed at B2 of chunk k.  (5) of chunk k is followed by (1), (2) of chunk
    // k + 1 without a barrier: different data.
    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): this wavefront's part of the first chunk has landed
    __syncthreads();
    GDML_TICK(0);                                       // prologue
    for (int m0 = 0, buf = 0; m0 < Mt; m0 += GDML_CH, buf = NB == 2 ? buf ^ 1 : 0) {
        const int mc = min(GDML_CH, Mt - m0);
        const double *sxs = L.stage + buf * 2 * GDML_CH * Dd, *sal = sxs + GDML_CH * Dd;
        if (NB == 2) {
            if (m0 + GDML_CH < Mt) stage_chunk(m0 + GDML_CH, buf ^ 1);
        } else if (m0 > 0) {
            __builtin_amdgcn_s_waitcnt(0x0F70);         // one buffer: this chunk was requested behind the previous chunk's
            __syncthreads();                            // last reader; everybody's part has landed
        }
        GDML_TICK(1);                                   // copy requests of the next chunk
        gather_begin();
        auto gathers = [&] { sfor<0, QN>([&](auto ccc) { gather_partner(ccc, sxs, sal); }); };
        // four wavefronts: gathers between the row reductions and the scalar tail (coumarin 3.50 -> 3.35 ms per stage launch);
        // eight wavefronts: after the tail (the interleaved order costs the larger kernel 28 more spilled registers)
        if (THREADS == 256) row_scalars(m0, sxs, sal, gathers);
        else { row_scalars(m0, sxs, sal, [] {}); GDML_TICK(2); gathers(); }
        gather_end();
        GDML_TICK(3);                                   // (2: row reductions + scalar tail) 3: gathers
        __syncthreads();
        GDML_TICK(4);                                   // barrier 1
        // (3) gradient terms of the chunk, (4) operand rows of the chunk
#ifndef GDML_ABLATE_GRAD
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int d = tid + nth * j;
            if (d < Dd) {
                // the exact rounding error of the addition (branch-free two-sum: the same value as the magnitude-ordered
                // form, without the compare and selects)
                // (round 4: all operands of a full chunk requested before the first term -- instead of two LDS round trips per
                // training point, one point after the other -- costs 35 more spilled registers and 4 % of the launch: the loop
                // stays rolled)
                auto add_term = [&](double t) {
                    // the exact rounding error of the addition (branch-free two-sum: the same value as the magnitude-ordered
                    // form, without the compare and selects)
                    const double sn = gacc[j] + t, bb = sn - gacc[j];
                    gcomp[j] += (gacc[j] - (sn - bb)) + (t - bb);
                    gacc[j] = sn;
                };
                int mm = 0;
                for (; mm + 1 < mc; mm += 2) {              // two training points per trip: eight LDS reads in flight instead of four
                    const double f0 = L.fm[mm], e0 = L.ea[mm], a0 = sal[mm * Dd + d], x0 = sxs[mm * Dd + d];
                    const double f1 = L.fm[mm + 1], e1 = L.ea[mm + 1], a1 = sal[(mm + 1) * Dd + d], x1 = sxs[(mm + 1) * Dd + d];
                    add_term(fma(f0, a0, -e0 * (xown[j] - x0)));
                    add_term(fma(f1, a1, -e1 * (xown[j] - x1)));
                }
                if (mm < mc) add_term(fma(L.fm[mm], sal[mm * Dd + d], -L.ea[mm] * (xown[j] - sxs[mm * Dd + d])));
            }
        }
#endif
        if (fm_active && fm_q == 0) {
#pragma unroll
            for (int i = 0; i < MPT; ++i) {
                const int mm = MPT * fm_mh + i;
                const bool in = mm < mc;                 // rows of a partial last chunk are zero
                const double w = in ? L.wm[mm] : 0.0, em = in ? L.em[mm] : 0.0;
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const double xj = in ? base[u] - sx[i][u] : 0.0, qv = -em * sa3[i][u];
                    L.P[mm * XP + 3 * fm_at + u] = xj; L.Qn[mm * XP + 3 * fm_at + u] = qv; L.Z[mm * XP + 3 * fm_at + u] = fma(w, xj, qv);
                }
            }
        }
        GDML_TICK(5);                                   // gradient terms + operand rows
        if (NB == 2) __builtin_amdgcn_s_waitcnt(0x0F70);   // this wavefront's part of chunk k + 1 has landed
        __syncthreads();
        GDML_TICK(6);                                   // copy wait + barrier 2
        if (NB == 1 && m0 + GDML_CH < Mt) stage_chunk(m0 + GDML_CH, 0);     // nobody reads the stage any more in this chunk
        // (5) [XJ ; -e AJ]^T [Z ; XJ] of the chunk on the matrix cores: all operands of the wavefront's tiles are requested
        //     first, then the MFMAs of the tiles run interleaved (one block, no wait between the products)
#ifndef GDML_ABLATE_MFMA
        {
            // tiles in groups of three: the operands of a group are requested first, then its MFMAs run interleaved
            constexpr int KS = GDML_CH / 4, TG = GDML_MAX_TILES > 3 ? 3 : GDML_MAX_TILES;
#pragma unroll
            for (int s0 = 0; s0 < GDML_MAX_TILES; s0 += TG) {
                double oa[TG][KS][2], ob[TG][KS][2];
#pragma unroll
                for (int g = 0; g < TG; ++g) {
                    const int sl = s0 + g < GDML_MAX_TILES ? s0 + g : GDML_MAX_TILES - 1;
                    const double *ar = L.P + rg * XP + 16 * tr_[sl] + li, *bc = L.Z + rg * XP + 16 * tc_[sl] + li;
                    const double *aq = L.Qn + rg * XP + 16 * tr_[sl] + li, *bp = L.P + rg * XP + 16 * tc_[sl] + li;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        oa[g][ks][0] = ar[4 * ks * XP]; ob[g][ks][0] = bc[4 * ks * XP];
                        oa[g][ks][1] = aq[4 * ks * XP]; ob[g][ks][1] = bp[4 * ks * XP];
                    }
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int w = 0; w < 2; ++w)
#pragma unroll
                        for (int g = 0; g < TG; ++g)
                            if (s0 + g < GDML_MAX_TILES)
                                acc[s0 + g] = __builtin_amdgcn_mfma_f64_16x16x4f64(oa[g][ks][w], ob[g][ks][w], acc[s0 + g], 0, 0, 0);
            }
        }
#endif
        GDML_TICK(7);                                   // matrix-core phase
    }
    double red2[2] = {esum, ssum};
    block_sum<2>(red2, L.red);
    const double energy = red2[0] * G.std + G.c, S = red2[1];
#pragma unroll
    for (int j = 0; j < EPT; ++j)
        if (tid + nth * j < Dd) L.gx[tid + nth * j] = gacc[j] + gcomp[j];
    __syncthreads();
    // ---- Cartesian gradient grad = std J^T g_x and the diagonal atom blocks of the pair terms,
    //      dg[a] = sum_c (-S jd jd^T + g d2x) = sum_c coef coef^T (3 g r - S) - 1 sum_c g x^3      (r = 1 / x, coef = -x^3 (r_a - r_c)),
    // from the Jacobian coefficients the formation threads hold for their partners: gathers like J^T xs_m, by the first
    // group of four threads per atom (each atom's 29 partners used to be walked by ONE thread per output element)
    if (tid < PARTS * N) {
        double g3[3] = {0.0, 0.0, 0.0}, h6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, tr3 = 0.0;
#pragma unroll
        for (int cc = 0; cc < QN; ++cc) {
            const int c = QN * fm_q + cc, d = pidx[cc];
            const bool ok = c < N && c != fm_at;             // coef = 0 otherwise; the trace term needs the mask
            const double x = L.x[d], g = ok ? L.gx[d] : 0.0, t = fma(3.0 * g, L.r[d], -S);
            tr3 = fma(g, x * x * x, tr3);
#pragma unroll
            for (int u = 0; u < 3; ++u) g3[u] = fma(coef[u][cc], g, g3[u]);
            const double c0 = coef[0][cc] * t, c1 = coef[1][cc] * t, c2 = coef[2][cc] * t;
            h6[0] = fma(c0, coef[0][cc], h6[0]); h6[1] = fma(c0, coef[1][cc], h6[1]); h6[2] = fma(c0, coef[2][cc], h6[2]);
            h6[3] = fma(c1, coef[1][cc], h6[3]); h6[4] = fma(c1, coef[2][cc], h6[4]); h6[5] = fma(c2, coef[2][cc], h6[5]);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) g3[u] = parts_sum(g3[u]);
#pragma unroll
        for (int i = 0; i < 6; ++i) h6[i] = parts_sum(h6[i]);
        tr3 = parts_sum(tr3);
        if (fm_q == 0) {
            double *dg = L.dg + 9 * fm_at;
#pragma unroll
            for (int u = 0; u < 3; ++u) L.grad[3 * fm_at + u] = g3[u] * G.std;
            dg[0] = h6[0] - tr3; dg[1] = h6[1]; dg[2] = h6[2];
            dg[3] = h6[1]; dg[4] = h6[3] - tr3; dg[5] = h6[4];
            dg[6] = h6[2]; dg[7] = h6[4]; dg[8] = h6[5] - tr3;
        }
    }
    __syncthreads();
    // atom-pair terms element by element, scale, store.  Round 3 stored every value twice from the accumulator layout -- the
    // mirror image by scalar stores with a stride of one matrix row: WRITE_SIZE 2.1 x the Hessian (profiles/r3_config5_hbm.json);
    // tiles stored in rows (mirror transposed through LDS) still write 1.56 x: a tile row is a 128-byte segment at an offset
    // of 16 r bytes from a 64-byte sector (rows are 720 B at 30 atoms), every segment ends in partial sectors.  Now the whole
    // symmetric matrix is assembled in LDS (the stage and operand buffers are dead) and streamed out linearly: whole sectors
    // except at the two ends of a matrix (profiles/r4_config5_hbm.json).  Molecules whose matrix does not fit the dead
    // buffers keep the tile stores.  The mirror holds the SAME values: the matrix is exactly symmetric.
    {
        const size_t room = (size_t)3 * GDML_CH * XP + (size_t)NB * 2 * GDML_CH * Dd;       // doubles from L.P to the end of the stage
        const bool image = (size_t)X * X + 1 <= room;
        double *img = L.P + (((uintptr_t)L.P >> 3) & 1);           // 16-byte aligned
        double *tbuf = L.P + wave * (16 * 17);                  // tile path: [16][17] doubles per wavefront
#pragma unroll
        for (int sl = 0; sl < GDML_MAX_TILES; ++sl) {
            if (wave + sl * nw >= ntiles) continue;
            const int y = 16 * tc_[sl] + li, b = y / 3, v = y - 3 * b;
            const bool diag_tile = tr_[sl] == tc_[sl];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int row = rg + 4 * qq, xr = 16 * tr_[sl] + row, a = xr / 3, u = xr - 3 * a;
                double val = 0.0;
                if (xr < X && y < X && !(diag_tile && xr > y)) {
                    double fin;
                    if (a == b) fin = L.dg[9 * a + 3 * u + v];
                    else {
                        const int d = pair_index(a, b);
                        const double x = L.x[d], g = L.gx[d], x3 = x * x * x, x5 = x3 * x * x;
                        const double du = L.pos[3 * a + u] - L.pos[3 * b + u], dv = L.pos[3 * a + v] - L.pos[3 * b + v];
                        fin = S * (x3 * du) * (x3 * dv) - (3.0 * g * x5 * du * dv - (u == v ? g * x3 : 0.0));
                    }
                    val = (acc[sl][qq] + fin) * G.std;
                    if (image) { img[xr * X + y] = val; img[y * X + xr] = val; }
                    // written once, read by the monodromy kernel after the stage: non-temporal, so that the 65 KB per geometry do
                    // not push the training set (read by every workgroup) out of the XCD's L2
                    else __builtin_nontemporal_store(val, &hess[(size_t)xr * X + y]);
                }
                if (!image) tbuf[row * 17 + li] = val;
            }
            if (!image) {
                wave_lds_fence();
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    // mirror element (row 16 tc + row, column 16 tr + li) = value (16 tr + li, 16 tc + row) of the tile
                    const int row = rg + 4 * qq, yy = 16 * tc_[sl] + row, xx = 16 * tr_[sl] + li;
                    const double val = tbuf[li * 17 + row];
                    if (yy < X && xx < X && (!diag_tile || li < row)) __builtin_nontemporal_store(val, &hess[(size_t)yy * X + xx]);
                }
                wave_lds_fence();
            }
        }
        if (image) {
            __syncthreads();
            for (int e = tid; e < X * X; e += nth) __builtin_nontemporal_store(img[e], &hess[e]);
        }
    }
    __syncthreads();
    GDML_TICK(8);                                       // epilogue: sums, Cartesian gradient, atom-pair terms, Hessian stores
#ifdef GDML_PHASE_CLOCK
    if (pc_on) { for (int i = 0; i < 10; ++i) g_phase_clock[i] += pc[i]; g_phase_clock[10] += 1; }
#endif
    return energy;
}

// ------------------------------------------------------------------ function-level evaluation
struct EvalArgs {
    sc_gdml_model G;
    const double *r;
    int64_t n;
    double *energy, *grad, *hess;
};

template <int THREADS, int HN, int CH, int MT, int NB>
__global__ __launch_bounds__(THREADS, 2) void gdml_eval_kernel(EvalArgs A) {
    extern __shared__ double smem[];
    const int X = 3 * A.G.n_atoms;
    const GdmlLds L = gdml_carve(smem, A.G.n_atoms, A.G.n_desc, CH);
    for (int64_t tr = blockIdx.x; tr < A.n; tr += gridDim.x) {
        __syncthreads();
        for (int i = threadIdx.x; i < X; i += blockDim.x) L.pos[i] = A.r[tr * X + i];
        __syncthreads();
        const double e = gdml_eval_device<HN, THREADS, CH, MT, NB>(A.G, L, A.hess + (size_t)tr * X * X);
        for (int i = threadIdx.x; i < X; i += blockDim.x) A.grad[tr * X + i] = L.grad[i];
        if (threadIdx.x == 0) A.energy[tr] = e - A.G.origin;
    }
}

// ------------------------------------------------------------------ RK4 stage of (q, p, S)
struct StageArgs {
    sc_gdml_model G;
    sc_state st;
    sc_dense_scratch sc;
    double dt;
    int stage;
    double *epart;
};

template <int THREADS, int HN, int CH, int MT, int NB>
__global__ __launch_bounds__(THREADS, 2) void gdml_stage_kernel(StageArgs A) {
    extern __shared__ double smem[];
    const int D = A.st.dim, tid = threadIdx.x, nth = blockDim.x, s = A.stage;
    const GdmlLds L = gdml_carve(smem, A.G.n_atoms, A.G.n_desc, CH);
    const double dt = A.dt, c = (s == 0) ? 0.0 : (s == 3 ? dt : 0.5 * dt), w = (s == 0 || s == 3) ? 1.0 : 2.0;
    const double h6 = dt / 6.0;
    double esum = 0.0;
    for (int64_t tr = blockIdx.x; tr < A.st.n; tr += gridDim.x) {
        double *qp = A.st.qp + tr * 2 * D;
        double *kprev = A.sc.kprev + tr * 2 * D, *ksum = A.sc.ksum + tr * 2 * D;
        __syncthreads();
        double ps[2] = {0.0, 0.0};   // up to 2 coordinates per thread (D <= 512)
        for (int i = tid, j = 0; i < D; i += nth, ++j) {
            const double kq = s ? kprev[i] : 0.0, kp = s ? kprev[D + i] : 0.0;
            L.pos[i] = qp[i] + c * kq;
            ps[j] = qp[D + i] + c * kp;
        }
        __syncthreads();
        const double e = gdml_eval_device<HN, THREADS, CH, MT, NB>(A.G, L, A.sc.hess + ((size_t)tr * 4 + s) * D * D) - A.G.origin;
        double tk[1] = {0.0};
        for (int i = tid, j = 0; i < D; i += nth, ++j) {
            const double im = A.G.inv_mass[i], kq = ps[j] * im, kp = -L.grad[i];
            tk[0] += 0.5 * ps[j] * ps[j] * im;
            kprev[i] = kq; kprev[D + i] = kp;
            const double sq = (s ? ksum[i] : 0.0) + w * kq, sp = (s ? ksum[D + i] : 0.0) + w * kp;
            if (s < 3) { ksum[i] = sq; ksum[D + i] = sp; }
            else { qp[i] += h6 * sq; qp[D + i] += h6 * sp; }
        }
        block_sum<1>(tk, L.red);
        if (tid == 0) {
            const double ds = tk[0] - e, acc = (s ? A.sc.ssum[tr] : 0.0) + w * ds;
            if (s < 3) A.sc.ssum[tr] = acc;
            else { A.st.act[tr] += h6 * acc; esum += tk[0] + e; }
        }
    }
    if (tid == 0 && A.epart && s == 3) A.epart[blockIdx.x] = esum;
}

}  // namespace

extern "C" int sc_gdml_row_len(int32_t n_atoms) {
    for (int len : {8, 16, 20, 24, 32, 40, 48})
        if (n_atoms <= len) return len;
    return -1;
}

namespace {

int check_model(const sc_gdml_model *g, const char *who) {
    if (!g || !g->xs_train || !g->jx_alphas || !g->pair_k || !g->pair_l)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "%s: null model field", who);
    if (g->n_atoms > 48) return sc_fail(SC_ERR_UNSUPPORTED, "%s: %d atoms (the instantiated kernels hold up to 48: four threads keep the "
                                        "Jacobian coefficients of an atom's partners in registers, a workgroup the chunk's rows in LDS)", who, g->n_atoms);
    if (((uintptr_t)g->xs_train | (uintptr_t)g->jx_alphas) & 15)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "%s: xs_train / jx_alphas must be 16-byte aligned", who);
    if (g->n_desc != g->n_atoms * (g->n_atoms - 1) / 2)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "%s: descriptor size %d does not match %d atoms", who, g->n_desc, g->n_atoms);
    if (gdml_lds_doubles(g->n_atoms, g->n_desc) * 8 > 160 * 1024)
        return sc_fail(SC_ERR_UNSUPPORTED, "%s: model (N=%d) needs more than 160 KiB of LDS", who, g->n_atoms);
    {
        const int T = (3 * g->n_atoms + 15) / 16;
        const GdmlShape sh = gdml_shape(g->n_atoms);
        if (T * (T + 1) / 2 > sh.mt * (sh.threads / 64))
            return sc_fail(SC_ERR_UNSUPPORTED, "%s: %d atoms need more Hessian tiles than the kernel holds", who, g->n_atoms);
    }
    return SC_OK;
}

}  // namespace

extern "C" int sc_gdml_eval(const sc_gdml_model *g, const double *r, int64_t n, double *energy, double *grad,
                            double *hess, void *stream) {
    int rc = check_model(g, "sc_gdml_eval");
    if (rc) return rc;
    if (!r || !energy || !grad || !hess) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_gdml_eval: null argument");
    if (n <= 0) return SC_OK;
    const size_t lds = gdml_lds_doubles(g->n_atoms, g->n_desc) * 8;
    EvalArgs a{*g, r, n, energy, grad, hess};
    const int grid = (int)(n < 1024 ? n : 1024);
#define SC_GDML_EVAL(TH_, HN_, CH_, MT_, NB_)                                                                                       \
    if (threads == TH_ && row_len == 2 * HN_ && ch == CH_ && nb == NB_) {                                                             \
        if (hipFuncSetAttribute((const void *)gdml_eval_kernel<TH_, HN_, CH_, MT_, NB_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds) != hipSuccess)                                                                   \
            return sc_check_launch("sc_gdml_eval (LDS attribute)");                                                        \
        hipLaunchKernelGGL((gdml_eval_kernel<TH_, HN_, CH_, MT_, NB_>), dim3(grid), dim3(TH_), lds, (hipStream_t)stream, a);         \
        launched = true;                                                                                                   \
    }
    const GdmlShape sh = gdml_shape(g->n_atoms);
    const int threads = sh.threads, row_len = sc_gdml_row_len(g->n_atoms), ch = sh.ch, nb = sh.nb;
    bool launched = false;
    SC_GDML_EVAL(256, 4, 4, 3, 2) SC_GDML_EVAL(256, 8, 4, 3, 2) SC_GDML_EVAL(256, 10, 4, 3, 2) SC_GDML_EVAL(256, 12, 4, 3, 2)
    SC_GDML_EVAL(512, 12, 8, 3, 2) SC_GDML_EVAL(512, 16, 8, 3, 2) SC_GDML_EVAL(512, 16, 4, 3, 2)
    SC_GDML_EVAL(512, 20, 4, 5, 2) SC_GDML_EVAL(512, 24, 4, 6, 1)
#undef SC_GDML_EVAL
    if (!launched) return sc_fail(SC_ERR_UNSUPPORTED, "sc_gdml_eval: no kernel for %d atoms", g->n_atoms);
    return sc_check_launch("sc_gdml_eval");
}

#ifdef GDML_PHASE_CLOCK
extern "C" int sc_gdml_phase_clock(unsigned long long *buf) {       // device buffer of 11 counters, or NULL to switch off
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase_clock), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#endif

extern "C" int sc_dense_grid(int64_t n) { return (int)(n < 512 ? (n > 0 ? n : 1) : 512); }

extern "C" int sc_gdml_stage(const sc_gdml_model *g, const sc_state *st, const sc_dense_scratch *sc, double dt,
                             int32_t stage, double *energy_partials, void *stream) {
    int rc = check_model(g, "sc_gdml_stage");
    if (rc) return rc;
    if (!st || !sc || !sc->hess || !sc->kprev || !sc->ksum || !sc->ssum)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_gdml_stage: null argument");
    if (st->dim != 3 * g->n_atoms) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_gdml_stage: dimension mismatch");
    if (stage < 0 || stage > 3) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_gdml_stage: stage %d", stage);
    if (st->dim > 512) return sc_fail(SC_ERR_UNSUPPORTED, "sc_gdml_stage: D=%d > 512", st->dim);
    if (st->n <= 0) return SC_OK;
    const size_t lds = gdml_lds_doubles(g->n_atoms, g->n_desc) * 8;
    StageArgs a{*g, *st, *sc, dt, stage, energy_partials};
#define SC_GDML_STAGE(TH_, HN_, CH_, MT_, NB_)                                                                                      \
    if (threads == TH_ && row_len == 2 * HN_ && ch == CH_ && nb == NB_) {                                                             \
        if (hipFuncSetAttribute((const void *)gdml_stage_kernel<TH_, HN_, CH_, MT_, NB_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds) != hipSuccess)                                                                   \
            return sc_check_launch("sc_gdml_stage (LDS attribute)");                                                       \
        hipLaunchKernelGGL((gdml_stage_kernel<TH_, HN_, CH_, MT_, NB_>), dim3(sc_dense_grid(st->n)), dim3(TH_), lds, (hipStream_t)stream, a); \
        launched = true;                                                                                                   \
    }
    const GdmlShape sh = gdml_shape(g->n_atoms);
    const int threads = sh.threads, row_len = sc_gdml_row_len(g->n_atoms), ch = sh.ch, nb = sh.nb;
    bool launched = false;
    SC_GDML_STAGE(256, 4, 4, 3, 2) SC_GDML_STAGE(256, 8, 4, 3, 2) SC_GDML_STAGE(256, 10, 4, 3, 2) SC_GDML_STAGE(256, 12, 4, 3, 2)
    SC_GDML_STAGE(512, 12, 8, 3, 2) SC_GDML_STAGE(512, 16, 8, 3, 2) SC_GDML_STAGE(512, 16, 4, 3, 2)
    SC_GDML_STAGE(512, 20, 4, 5, 2) SC_GDML_STAGE(512, 24, 4, 6, 1)
#undef SC_GDML_STAGE
    if (!launched) return sc_fail(SC_ERR_UNSUPPORTED, "sc_gdml_stage: no kernel for %d atoms", g->n_atoms);
    return sc_check_launch("sc_gdml_stage");
}
