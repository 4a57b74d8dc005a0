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

namespace {

__device__ __forceinline__ int pair_index(int a, int b) {      // a != b; torch.tril_indices order (i > j)
    return a > b ? a * (a - 1) / 2 + b : b * (b - 1) / 2 + a;
}

struct GdmlLds {
    double *pos, *x, *jd, *gx, *fm, *em, *wm, *ea, *grad, *P, *Qn, *Z, *dg, *xsL, *aL, *red;
    int XP;      // row stride of the MFMA operand arrays P = XJ, Qn = -e AJ, Z = w XJ - e AJ  ([chunk][XP])
};

#define GDML_MAX_TILES 6      // 16 x 16 accumulator tiles per wavefront (21 tiles of a 30-atom molecule on 4 wavefronts)

// row stride of the operand arrays: 16 T (+16) doubles with stride = 16 mod 32, so that the four rows an MFMA operand
// read touches (64 lanes x 8 bytes) fall into different LDS banks
__host__ __device__ inline int gdml_xp(int N) {
    const int T = (3 * N + 15) / 16;
    return (T & 1) ? 16 * T : 16 * T + 16;
}

#define GDML_CHUNK_MAX 16     // training points staged per chunk (run-time choice: 16, 8 or 4, by the LDS budget)

__device__ GdmlLds gdml_carve(double *base, int N, int Dd, int Mt, int chunk) {
    GdmlLds L;
    double *f = base;
    L.red = f;  f += 32;
    L.pos = f;  f += 3 * N;
    L.x = f;    f += Dd;
    L.jd = f;   f += 3 * Dd;
    L.gx = f;   f += Dd;
    L.fm = f;   f += Mt;
    L.em = f;   f += Mt;
    L.wm = f;   f += Mt;
    L.ea = f;   f += Mt;
    L.grad = f; f += 3 * N + (N & 1);
    L.dg = f;   f += 9 * N + (N & 1);
    L.XP = gdml_xp(N);
    L.P = f;                           // operand arrays of the Hessian products; before that phase the same
    L.Qn = f + chunk * L.XP;           // storage holds the per-wavefront partial sums of the descriptor gradient
    L.Z = f + 2 * chunk * L.XP;        // (8 rows of Dd)
    L.xsL = f;
    L.aL = f;
    return L;
}

size_t gdml_lds_doubles(int N, int Dd, int Mt, int chunk) {
    return 32 + 3 * N + Dd + 3 * Dd + Dd + 4 * (size_t)Mt + 3 * N + (N & 1) + 9 * N + (N & 1) +
           (3 * (size_t)chunk * gdml_xp(N) > 8 * (size_t)Dd ? 3 * (size_t)chunk * gdml_xp(N) : 8 * (size_t)Dd);
}

// V (without origin), grad[3N] (LDS, L.grad) and hess[3N][3N] (global, row-major) at the geometry in L.pos.
// Every thread returns the energy.
// HN = partner atoms per half row of the square-form training data (rows of 2 HN doubles, sc_gdml_model.row_len)
template <int HN>
__device__ double gdml_eval_device(const sc_gdml_model &G, const GdmlLds &L, double *hess, int chunk) {
    const int N = G.n_atoms, Dd = G.n_desc, Mt = G.n_train, X = 3 * N;
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nth >> 6;
    const double q = G.q;
    // ---- descriptor and Jacobian rows
    for (int d = tid; d < Dd; d += nth) {
        const int k = G.pair_k[d], l = G.pair_l[d];
        const double dx = L.pos[3 * k] - L.pos[3 * l], dy = L.pos[3 * k + 1] - L.pos[3 * l + 1],
                     dz = L.pos[3 * k + 2] - L.pos[3 * l + 2];
        const double x = 1.0 / sqrt(dx * dx + dy * dy + dz * dz), x3 = x * x * x;
        L.x[d] = x;
        L.jd[3 * d] = -x3 * dx; L.jd[3 * d + 1] = -x3 * dy; L.jd[3 * d + 2] = -x3 * dz;
    }
    __syncthreads();
    // ---- per-training-point scalars (one wave per m, lanes over the descriptor)
    double esum = 0.0, ssum = 0.0;
    // four training points per wavefront and pass: their 8 row loads are in flight together and lanes 0..3 do the
    // scalar tail (sqrt, exp) of one point each
    for (int mq = 4 * wave; mq < Mt; mq += 4 * nw) {
        double s2[4] = {0, 0, 0, 0}, sa[4] = {0, 0, 0, 0};
        for (int d = lane; d < Dd; d += 64) {
            const double xv = L.x[d];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const size_t row = (size_t)min(mq + i, Mt - 1) * Dd + d;
                const double xd = xv - G.xs_train[row];
                s2[i] = fma(xd, xd, s2[i]);
                sa[i] = fma(xd, G.jx_alphas[row], sa[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { s2[i] = wave_sum(s2[i]); sa[i] = wave_sum(sa[i]); }
        const int m = mq + lane;
        if (lane < 4 && m < Mt) {
            const double s2m = lane == 0 ? s2[0] : lane == 1 ? s2[1] : lane == 2 ? s2[2] : s2[3];
            const double sam = lane == 0 ? sa[0] : lane == 1 ? sa[1] : lane == 2 ? sa[2] : sa[3];
            const double dist = sqrt(s2m), e = (1.0 / 3.0) * q * q * q * q * exp(-q * dist);
            const double f = e * (1.0 + q * dist) / (q * q);
            L.fm[m] = f; L.em[m] = e; L.wm[m] = e * sam * q / dist; L.ea[m] = e * sam;
            esum += f * sam; ssum += e * sam;
        }
    }
    double red2[2] = {esum, ssum};
    block_sum<2>(red2, L.red);
    const double energy = red2[0] * G.std + G.c, S = red2[1];
    __syncthreads();
    // ---- gradient in descriptor space, then Cartesian gradient
    // every wavefront takes a slice of the training points (lanes over the descriptor: coalesced rows), the per-wave
    // partial sums meet in LDS (L.xsL = the storage of the Hessian operand arrays, not yet in use: one row of Dd per wavefront)
    for (int d0 = 0; d0 < Dd; d0 += 64) {
        const int d = d0 + lane;
        if (d < Dd) {
            // the terms (2e8) cancel to the size of the force (6e1): Neumaier-compensated accumulation keeps the
            // rounding of the SUM out of the result (what remains is the rounding of the terms themselves)
            const double xv = L.x[d];
            double g = 0.0, comp = 0.0;
            for (int m = wave; m < Mt; m += nw) {
                const double t = fma(L.fm[m], G.jx_alphas[(size_t)m * Dd + d], -L.ea[m] * (xv - G.xs_train[(size_t)m * Dd + d]));
                const double s = g + t;
                comp += fabs(g) >= fabs(t) ? (g - s) + t : (t - s) + g;
                g = s;
            }
            L.xsL[wave * Dd + d] = g + comp;
        }
    }
    __syncthreads();
    for (int d = tid; d < Dd; d += nth) {
        double g = 0.0, comp = 0.0;
        for (int w = 0; w < nw; ++w) {
            const double t = L.xsL[w * Dd + d];
            const double s = g + t;
            comp += fabs(g) >= fabs(t) ? (g - s) + t : (t - s) + g;
            g = s;
        }
        L.gx[d] = g + comp;
    }
    __syncthreads();
    for (int xi = tid; xi < X; xi += nth) {
        const int a = xi / 3, u = xi - 3 * a;
        double g = 0.0;
        for (int b = 0; b < N; ++b) {
            if (b == a) continue;
            const int d = pair_index(a, b);
            const double j = L.jd[3 * d + u] * L.gx[d];
            g += (a > b) ? j : -j;
        }
        L.grad[xi] = g * G.std;
    }
    // ---- Hessian.  Diagonal atom blocks of the pair terms first: dg[a] = sum_c (-S jd jd^T + d2x) over the partners
    for (int e = tid; e < 9 * N; e += nth) {
        const int a = e / 9, u = (e - 9 * a) / 3, v = e - 9 * a - 3 * u;
        double acc = 0.0;
        for (int c = 0; c < N; ++c) {
            if (c == a) continue;
            const int d = pair_index(a, c);
            const double x = L.x[d], g = L.gx[d], x3 = x * x * x, x5 = x3 * x * x, ix3 = -1.0 / x3;
            const double ju = L.jd[3 * d + u], jv = L.jd[3 * d + v];
            acc += -S * ju * jv + 3.0 * g * x5 * (ju * ix3) * (jv * ix3) - (u == v ? g * x3 : 0.0);
        }
        L.dg[e] = acc;
    }
    // rank-M sums on the matrix cores.  Tile t = c (c + 1) / 2 + r, r <= c, of the upper triangle belongs to wavefront
    // t % nw; accumulator layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 reg.
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int XP = L.XP, T = (X + 15) / 16, ntiles = T * (T + 1) / 2, rg = lane >> 4, li = lane & 15;
    d4 acc[GDML_MAX_TILES];
    int tr_[GDML_MAX_TILES], tc_[GDML_MAX_TILES];
#pragma unroll
    for (int sl = 0; sl < GDML_MAX_TILES; ++sl) {
        acc[sl] = (d4){0.0, 0.0, 0.0, 0.0};
        const int t = wave + sl * nw;
        int c = 0;
        while ((c + 1) * (c + 2) / 2 <= t) ++c;
        tr_[sl] = t - c * (c + 1) / 2; tc_[sl] = c;
    }
    for (int e = tid; e < 3 * chunk * XP; e += nth) L.P[e] = 0.0;       // P, Qn, Z are contiguous: padding columns stay 0
    // formation roles: thread = (training point slot, atom, half of the partner atoms)
    const int fm_rows = nth / (2 * N) < chunk ? nth / (2 * N) : chunk;      // training points formed per pass
    const int fm_half = tid & 1, fm_at = (tid >> 1) % N, fm_mm = (tid >> 1) / N;
    const bool fm_active = fm_mm < fm_rows;
    double coef[3][HN], base[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int cc = 0; cc < HN; ++cc) {
        const int c = HN * fm_half + cc;
        const bool ok = c < N && c != fm_at;
        const int d = ok ? pair_index(fm_at, c) : 0;
        const double xq = ok ? L.x[d] : 0.0;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const double j = ok ? ((fm_at > c) ? L.jd[3 * d + u] : -L.jd[3 * d + u]) : 0.0;
            coef[u][cc] = j;
            base[u] = fma(j, xq, base[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) base[u] += dpp_mov_f64<0xB1>(base[u]);    // both halves of the partner sum
    for (int m0 = 0; m0 < Mt; m0 += chunk) {
        const int mc = min(chunk, Mt - m0);
        __syncthreads();
        // XJ_m = J^T xd_m, AJ_m = J^T A_m for the chunk, from the SQUARE form of the training data
        //   xs_sq[m][a][c] = xs_m[pair(a, c)],  a_sq[m][a][c] = A_m[pair(a, c)]   (rows of 32 doubles, zero for c = a, c >= N)
        // read straight from L2 (the rows of one atom are contiguous): a thread pair (half = 0, 1) owns (atom a, training
        // point mm), each half takes 16 partner atoms with its coefficients coef[u][c] = +-jd[pair(a, c)][u] in
        // registers (loaded once per geometry), so a training value feeds three multiply-adds and nothing is gathered:
        //   XJ_m[a, u] = base[u] - sum_c coef[u][c] xs_sq[m][a][c],   AJ_m[a, u] = sum_c coef[u][c] a_sq[m][a][c]
        if (mc < chunk)                                  // last, partial chunk: its unused operand rows must be zero
            for (int e = mc * XP + tid; e < chunk * XP; e += nth) { L.P[e] = 0.0; L.Qn[e] = 0.0; L.Z[e] = 0.0; }
#ifndef GDML_ABLATE_FORM
        for (int pass = 0; pass < chunk; pass += fm_rows) {
            const int mm = pass + fm_mm;
            double sx[3] = {0.0, 0.0, 0.0}, sa[3] = {0.0, 0.0, 0.0};
            if (fm_active && mm < mc) {
                const size_t row = (((size_t)(m0 + mm) * N + fm_at) * (2 * HN) + HN * fm_half);
                const double2 *xr = (const double2 *)(G.xs_sq + row), *ar = (const double2 *)(G.a_sq + row);
                double2 xv[HN / 2], av[HN / 2];
#pragma unroll
                for (int i = 0; i < HN / 2; ++i) { xv[i] = xr[i]; av[i] = ar[i]; }
#pragma unroll
                for (int i = 0; i < HN / 2; ++i) {
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        sx[u] = fma(coef[u][2 * i], xv[i].x, sx[u]); sx[u] = fma(coef[u][2 * i + 1], xv[i].y, sx[u]);
                        sa[u] = fma(coef[u][2 * i], av[i].x, sa[u]); sa[u] = fma(coef[u][2 * i + 1], av[i].y, sa[u]);
                    }
                }
            }
            // the two halves of a row sit in adjacent lanes
#pragma unroll
            for (int u = 0; u < 3; ++u) { sx[u] += dpp_mov_f64<0xB1>(sx[u]); sa[u] += dpp_mov_f64<0xB1>(sa[u]); }   // quad_perm [1,0,3,2]
            if (fm_active && fm_half == 0 && mm < mc) {
                const double w = L.wm[m0 + mm], em = L.em[m0 + mm];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const double xj = base[u] - sx[u], q = -em * sa[u];
                    L.P[mm * XP + 3 * fm_at + u] = xj; L.Qn[mm * XP + 3 * fm_at + u] = q; L.Z[mm * XP + 3 * fm_at + u] = fma(w, xj, q);
                }
            }
        }
#endif
        __syncthreads();
#ifndef GDML_ABLATE_MFMA
#pragma unroll
        for (int sl = 0; sl < GDML_MAX_TILES; ++sl) {
            if (wave + sl * nw >= ntiles) continue;               // wave-uniform
            const double *ar = L.P + rg * XP + 16 * tr_[sl] + li, *bc = L.Z + rg * XP + 16 * tc_[sl] + li;
            const double *aq = L.Qn + rg * XP + 16 * tr_[sl] + li, *bp = L.P + rg * XP + 16 * tc_[sl] + li;
            for (int ks = 0; ks < chunk / 4; ++ks) {
                acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[4 * ks * XP], bc[4 * ks * XP], acc[sl], 0, 0, 0);
                acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[4 * ks * XP], bp[4 * ks * XP], acc[sl], 0, 0, 0);
            }
        }
#endif
    }
    __syncthreads();         // dg complete (written before the chunk loop's first barrier anyway)
    // atom-pair terms element by element, scale, write both triangles
#pragma unroll
    for (int sl = 0; sl < GDML_MAX_TILES; ++sl) {
        if (wave + sl * nw >= ntiles) continue;
        const int y = 16 * tc_[sl] + li, b = y / 3, v = y - 3 * b;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int xr = 16 * tr_[sl] + rg + 4 * q, a = xr / 3, u = xr - 3 * a;
            if (xr >= X || y >= X || (tr_[sl] == tc_[sl] && xr > y)) continue;
            double fin;
            if (a == b) fin = L.dg[9 * a + 3 * u + v];
            else {
                const int d = pair_index(a, b);
                const double x = L.x[d], g = L.gx[d], x3 = x * x * x, x5 = x3 * x * x, ix3 = -1.0 / x3;
                const double ju = L.jd[3 * d + u], jv = L.jd[3 * d + v];
                fin = S * ju * jv - (3.0 * g * x5 * (ju * ix3) * (jv * ix3) - (u == v ? g * x3 : 0.0));
            }
            const double val = (acc[sl][q] + fin) * G.std;
            hess[(size_t)xr * X + y] = val;
            hess[(size_t)y * X + xr] = val;
        }
    }
    __syncthreads();
    return energy;
}

// ------------------------------------------------------------------ function-level evaluation
struct EvalArgs {
    sc_gdml_model G;
    int chunk;
    const double *r;
    int64_t n;
    double *energy, *grad, *hess;
};

template <int THREADS, int HN>
__global__ __launch_bounds__(THREADS, 512 / THREADS) void gdml_eval_kernel(EvalArgs A) {
    extern __shared__ double smem[];
    const int X = 3 * A.G.n_atoms;
    const GdmlLds L = gdml_carve(smem, A.G.n_atoms, A.G.n_desc, A.G.n_train, A.chunk);
    for (int64_t tr = blockIdx.x; tr < A.n; tr += gridDim.x) {
        __syncthreads();
        for (int i = threadIdx.x; i < X; i += blockDim.x) L.pos[i] = A.r[tr * X + i];
        __syncthreads();
        const double e = gdml_eval_device<HN>(A.G, L, A.hess + (size_t)tr * X * X, A.chunk);
        for (int i = threadIdx.x; i < X; i += blockDim.x) A.grad[tr * X + i] = L.grad[i];
        if (threadIdx.x == 0) A.energy[tr] = e - A.G.origin;
    }
}

// ------------------------------------------------------------------ RK4 stage of (q, p, S)
struct StageArgs {
    sc_gdml_model G;
    int chunk;
    sc_state st;
    sc_dense_scratch sc;
    double dt;
    int stage;
    double *epart;
};

template <int THREADS, int HN>
__global__ __launch_bounds__(THREADS, 512 / THREADS) void gdml_stage_kernel(StageArgs A) {
    extern __shared__ double smem[];
    const int D = A.st.dim, tid = threadIdx.x, nth = blockDim.x, s = A.stage;
    const GdmlLds L = gdml_carve(smem, A.G.n_atoms, A.G.n_desc, A.G.n_train, A.chunk);
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
        const double e = gdml_eval_device<HN>(A.G, L, A.sc.hess + ((size_t)tr * 4 + s) * D * D, A.chunk) - A.G.origin;
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

// training points per staged chunk: the largest of 16, 8, 4 that fits LDS (measured at 30 atoms: a smaller chunk that
// lets two workgroups share a CU is slower -- more chunk iterations, each with three barriers); the per-wave partial
}  // namespace

extern "C" int sc_gdml_row_len(int32_t n_atoms) {
    for (int len : {8, 16, 20, 24, 32})
        if (n_atoms <= len) return len;
    return -1;
}

namespace {

int gdml_chunk(const sc_gdml_model *g, int threads) {
    (void)threads;
    const int minc = 4;
    for (int c = GDML_CHUNK_MAX; c >= minc; c /= 2)
        if (gdml_lds_doubles(g->n_atoms, g->n_desc, g->n_train, c) * 8 <= 160 * 1024) return c;
    return 0;
}

int check_model(const sc_gdml_model *g, const char *who) {
    if (!g || !g->xs_train || !g->jx_alphas || !g->pair_k || !g->pair_l)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "%s: null model field", who);
    if (!g->xs_sq || !g->a_sq) return sc_fail(SC_ERR_BAD_ARGUMENT, "%s: null square-form training arrays", who);
    if (g->n_atoms > 32) return sc_fail(SC_ERR_UNSUPPORTED, "%s: %d atoms (the square-form rows hold 32)", who, g->n_atoms);
    if (g->row_len != sc_gdml_row_len(g->n_atoms))
        return sc_fail(SC_ERR_BAD_ARGUMENT, "%s: row_len %d, %d atoms need %d", who, g->row_len, g->n_atoms, sc_gdml_row_len(g->n_atoms));
    if (g->n_desc != g->n_atoms * (g->n_atoms - 1) / 2)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "%s: descriptor size %d does not match %d atoms", who, g->n_desc, g->n_atoms);
    if (gdml_chunk(g, 256) == 0)
        return sc_fail(SC_ERR_UNSUPPORTED, "%s: model (N=%d, M=%d) needs more than 160 KiB of LDS", who, g->n_atoms, g->n_train);
    {
        const int T = (3 * g->n_atoms + 15) / 16, nw = 4;
        if (T * (T + 1) / 2 > GDML_MAX_TILES * nw)
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
    const int chunk = gdml_chunk(g, 256);          // four wavefronts per geometry: two geometries share a CU
    const size_t lds = gdml_lds_doubles(g->n_atoms, g->n_desc, g->n_train, chunk) * 8;
    EvalArgs a{*g, chunk, r, n, energy, grad, hess};
    const int grid = (int)(n < 1024 ? n : 1024);
#define SC_GDML_EVAL(HN_)                                                                                                  \
    case 2 * HN_:                                                                                                          \
        if (hipFuncSetAttribute((const void *)gdml_eval_kernel<256, HN_>, hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                (int)lds) != hipSuccess)                                                                   \
            return sc_check_launch("sc_gdml_eval (LDS attribute)");                                                        \
        hipLaunchKernelGGL((gdml_eval_kernel<256, HN_>), dim3(grid), dim3(256), lds, (hipStream_t)stream, a);              \
        break;
    switch (g->row_len) { SC_GDML_EVAL(4) SC_GDML_EVAL(8) SC_GDML_EVAL(10) SC_GDML_EVAL(12) SC_GDML_EVAL(16) }
#undef SC_GDML_EVAL
    return sc_check_launch("sc_gdml_eval");
}

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
    const int chunk = gdml_chunk(g, 256);
    const size_t lds = gdml_lds_doubles(g->n_atoms, g->n_desc, g->n_train, chunk) * 8;
    StageArgs a{*g, chunk, *st, *sc, dt, stage, energy_partials};
#define SC_GDML_STAGE(HN_)                                                                                                 \
    case 2 * HN_:                                                                                                          \
        if (hipFuncSetAttribute((const void *)gdml_stage_kernel<256, HN_>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                (int)lds) != hipSuccess)                                                                   \
            return sc_check_launch("sc_gdml_stage (LDS attribute)");                                                       \
        hipLaunchKernelGGL((gdml_stage_kernel<256, HN_>), dim3(sc_dense_grid(st->n)), dim3(256), lds, (hipStream_t)stream, a); \
        break;
    switch (g->row_len) { SC_GDML_STAGE(4) SC_GDML_STAGE(8) SC_GDML_STAGE(10) SC_GDML_STAGE(12) SC_GDML_STAGE(16) }
#undef SC_GDML_STAGE
    return sc_check_launch("sc_gdml_stage");
}
