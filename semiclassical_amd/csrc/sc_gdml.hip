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
//   J^T J and the second-derivative term are 3x3 blocks per atom pair.  Everything is organised by atom-pair blocks.
#include "sc_common.h"
#include "sc_prefactor.h"

namespace {

__device__ __forceinline__ int pair_index(int a, int b) {      // a != b; torch.tril_indices order (i > j)
    return a > b ? a * (a - 1) / 2 + b : b * (b - 1) / 2 + a;
}

struct GdmlLds {
    double *pos, *x, *jd, *gx, *fm, *em, *wm, *ea, *grad, *XJ, *AJ, *xsL, *aL, *red;
};

#define GDML_CHUNK_MAX 16     // training points staged per chunk (run-time choice: 16, 8 or 4, by the LDS budget)
#define GDML_NB 2            // atom-pair blocks of the Hessian per thread and group

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
    L.grad = f; f += 3 * N;
    L.XJ = f;   f += chunk * 3 * N;
    L.AJ = f;   f += chunk * 3 * N;
    L.xsL = f;  f += chunk * Dd;       // training descriptors / coefficients of the chunk: staged once (coalesced),
    L.aL = f;   f += chunk * Dd;       // then read 2 (N-1) times each by the J^T products
    return L;
}

size_t gdml_lds_doubles(int N, int Dd, int Mt, int chunk) {
    return 32 + 3 * N + Dd + 3 * Dd + Dd + 4 * (size_t)Mt + 3 * N + 2 * (size_t)chunk * 3 * N +
           2 * (size_t)chunk * Dd;
}

// V (without origin), grad[3N] (LDS, L.grad) and hess[3N][3N] (global, row-major) at the geometry in L.pos.
// Every thread returns the energy.
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
    // partial sums meet in LDS (L.xsL and L.aL are contiguous and free at this point: 2 * chunk >= 2 * nw rows of Dd)
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
            L.xsL[wave * Dd + d] = g;
            L.xsL[(nw + wave) * Dd + d] = comp;
        }
    }
    __syncthreads();
    for (int d = tid; d < Dd; d += nth) {
        double g = 0.0, comp = 0.0;
        for (int w = 0; w < nw; ++w) {
            const double t = L.xsL[w * Dd + d];
            const double s = g + t;
            comp += (fabs(g) >= fabs(t) ? (g - s) + t : (t - s) + g) + L.xsL[(nw + w) * Dd + d];
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
    // ---- Hessian by atom-pair blocks (a <= b).  A thread owns up to GDML_NB blocks of a group, so that the chunk
    // products J^T xd_m, J^T A_m are formed once per group of GDML_NB * blockDim blocks (once in all for N <= 31).
    const int nblk = N * (N + 1) / 2;
    auto finish_block = [&](double (&h)[3][3], int a, int b) {
        // - S J^T J + second derivatives of the descriptor
        if (a != b) {
            const int d = pair_index(a, b);          // b > a: b is the "k" atom of the pair
            const double x = L.x[d], g = L.gx[d];
            const double jx = L.jd[3 * d], jy = L.jd[3 * d + 1], jz = L.jd[3 * d + 2];
            const double jv[3] = {jx, jy, jz};
            // diff = r_k - r_l = -jd / x^3
            const double ix3 = -1.0 / (x * x * x);
            const double df[3] = {jx * ix3, jy * ix3, jz * ix3};
            const double x5 = x * x * x * x * x, x3 = x * x * x;
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int v = 0; v < 3; ++v) {
                    const double T = 3.0 * g * x5 * df[u] * df[v] - (u == v ? g * x3 : 0.0);
                    h[u][v] += S * jv[u] * jv[v] - T;          // J^T J block is -jd jd^T
                }
        } else {
            for (int c = 0; c < N; ++c) {
                if (c == a) continue;
                const int d = pair_index(a, c);
                const double x = L.x[d], g = L.gx[d];
                const double jv[3] = {L.jd[3 * d], L.jd[3 * d + 1], L.jd[3 * d + 2]};
                const double ix3 = -1.0 / (x * x * x);
                const double df[3] = {jv[0] * ix3, jv[1] * ix3, jv[2] * ix3};
                const double x5 = x * x * x * x * x, x3 = x * x * x;
#pragma unroll
                for (int u = 0; u < 3; ++u)
#pragma unroll
                    for (int v = 0; v < 3; ++v) {
                        const double T = 3.0 * g * x5 * df[u] * df[v] - (u == v ? g * x3 : 0.0);
                        h[u][v] += -S * jv[u] * jv[v] + T;
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const double val = h[u][v] * G.std;
                hess[(size_t)(3 * a + u) * X + 3 * b + v] = val;
                hess[(size_t)(3 * b + v) * X + 3 * a + u] = val;
            }
    };
    for (int blk0 = 0; blk0 < nblk; blk0 += GDML_NB * nth) {
        int ba[GDML_NB], bb[GDML_NB];
        bool own[GDML_NB];
        double h[GDML_NB][3][3];
#pragma unroll
        for (int sl = 0; sl < GDML_NB; ++sl) {
            const int blk = blk0 + sl * nth + tid;
            own[sl] = blk < nblk;
            int a = 0, b = 0;
            if (own[sl]) {   // blk = b (b+1)/2 + a with a <= b
                b = (int)((sqrt(8.0 * blk + 1.0) - 1.0) * 0.5);
                while (b * (b + 1) / 2 > blk) --b;
                while ((b + 1) * (b + 2) / 2 <= blk) ++b;
                a = blk - b * (b + 1) / 2;
            }
            ba[sl] = a; bb[sl] = b;
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int v = 0; v < 3; ++v) h[sl][u][v] = 0.0;
        }
        for (int m0 = 0; m0 < Mt; m0 += chunk) {
            const int mc = min(chunk, Mt - m0);
            __syncthreads();
            // the chunk's training rows m0 .. m0+mc-1 are contiguous: coalesced global -> LDS.  (Read straight from L2
            // by the J^T products they cost 8 scattered bytes per multiply-add: the launch was L2-bandwidth bound.)
            for (int e = tid; e < mc * Dd; e += nth) {
                L.xsL[e] = G.xs_train[(size_t)m0 * Dd + e];
                L.aL[e] = G.jx_alphas[(size_t)m0 * Dd + e];
            }
            __syncthreads();
            // XJ_m = J^T xd_m, AJ_m = J^T A_m for the chunk.  Thread = (Cartesian component xi, group g of training
            // points mm = g, g + NG, ...): the Jacobian entry of a partner atom is fetched once per group, not per m.
            {
                const int NG = nth / X > 0 ? nth / X : 1;             // groups of training points
                constexpr int MPT = 4;                                // training points per thread and pass
                if (tid < NG * X) {
                    const int xi = tid % X, g = tid / X, at = xi / 3, u = xi - 3 * at;
                    for (int mb = g; mb < mc; mb += NG * MPT) {
                        double sx[MPT], sa[MPT];
#pragma unroll
                        for (int i = 0; i < MPT; ++i) { sx[i] = 0.0; sa[i] = 0.0; }
                        for (int c = 0; c < N; ++c) {
                            if (c == at) continue;
                            const int d = pair_index(at, c);
                            const double j = (at > c) ? L.jd[3 * d + u] : -L.jd[3 * d + u];
                            const double xd = L.x[d];
#pragma unroll
                            for (int i = 0; i < MPT; ++i) {
                                const int mm = mb + i * NG;
                                if (mm < mc) {
                                    sx[i] = fma(j, xd - L.xsL[mm * Dd + d], sx[i]);
                                    sa[i] = fma(j, L.aL[mm * Dd + d], sa[i]);
                                }
                            }
                        }
#pragma unroll
                        for (int i = 0; i < MPT; ++i) {
                            const int mm = mb + i * NG;
                            if (mm < mc) { L.XJ[mm * X + xi] = sx[i]; L.AJ[mm * X + xi] = sa[i]; }
                        }
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int sl = 0; sl < GDML_NB; ++sl) {
                if (!own[sl]) continue;
                const int a = ba[sl], b = bb[sl];
                for (int mm = 0; mm < mc; ++mm) {
                    const double w = L.wm[m0 + mm], e = L.em[m0 + mm];
                    const double *xj = L.XJ + mm * X, *aj = L.AJ + mm * X;
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const double xa = xj[3 * a + u], aa = aj[3 * a + u];
#pragma unroll
                        for (int v = 0; v < 3; ++v) {
                            const double xb = xj[3 * b + v], ab = aj[3 * b + v];
                            h[sl][u][v] += w * xa * xb - e * (aa * xb + xa * ab);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int sl = 0; sl < GDML_NB; ++sl)
            if (own[sl]) finish_block(h[sl], ba[sl], bb[sl]);
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

__global__ __launch_bounds__(256) void gdml_eval_kernel(EvalArgs A) {
    extern __shared__ double smem[];
    const int X = 3 * A.G.n_atoms;
    const GdmlLds L = gdml_carve(smem, A.G.n_atoms, A.G.n_desc, A.G.n_train, A.chunk);
    for (int64_t tr = blockIdx.x; tr < A.n; tr += gridDim.x) {
        __syncthreads();
        for (int i = threadIdx.x; i < X; i += blockDim.x) L.pos[i] = A.r[tr * X + i];
        __syncthreads();
        const double e = gdml_eval_device(A.G, L, A.hess + (size_t)tr * X * X, A.chunk);
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

template <int THREADS>
__global__ __launch_bounds__(THREADS) void gdml_stage_kernel(StageArgs A) {
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
        const double e = gdml_eval_device(A.G, L, A.sc.hess + ((size_t)tr * 4 + s) * D * D, A.chunk) - A.G.origin;
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
// sums (value + compensation term) of the descriptor gradient need chunk >= wavefronts
int gdml_chunk(const sc_gdml_model *g, int threads) {
    const int minc = threads / 64 > 4 ? threads / 64 : 4;
    for (int c = GDML_CHUNK_MAX; c >= minc; c /= 2)
        if (gdml_lds_doubles(g->n_atoms, g->n_desc, g->n_train, c) * 8 <= 160 * 1024) return c;
    return 0;
}

int check_model(const sc_gdml_model *g, const char *who) {
    if (!g || !g->xs_train || !g->jx_alphas || !g->pair_k || !g->pair_l)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "%s: null model field", who);
    if (g->n_desc != g->n_atoms * (g->n_atoms - 1) / 2)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "%s: descriptor size %d does not match %d atoms", who, g->n_desc, g->n_atoms);
    if (gdml_chunk(g, 512) == 0)
        return sc_fail(SC_ERR_UNSUPPORTED, "%s: model (N=%d, M=%d) needs more than 160 KiB of LDS", who, g->n_atoms, g->n_train);
    return SC_OK;
}

}  // namespace

extern "C" int sc_gdml_eval(const sc_gdml_model *g, const double *r, int64_t n, double *energy, double *grad,
                            double *hess, void *stream) {
    int rc = check_model(g, "sc_gdml_eval");
    if (rc) return rc;
    if (!r || !energy || !grad || !hess) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_gdml_eval: null argument");
    if (n <= 0) return SC_OK;
    const int chunk = gdml_chunk(g, 256);
    const size_t lds = gdml_lds_doubles(g->n_atoms, g->n_desc, g->n_train, chunk) * 8;
    EvalArgs a{*g, chunk, r, n, energy, grad, hess};
    if (hipFuncSetAttribute((const void *)gdml_eval_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return sc_check_launch("sc_gdml_eval (LDS attribute)");
    const int grid = (int)(n < 1024 ? n : 1024);
    hipLaunchKernelGGL(gdml_eval_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
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
    const bool big = g->n_atoms * (g->n_atoms + 1) / 2 > 256;
    const int chunk = gdml_chunk(g, big ? 512 : 256);
    const size_t lds = gdml_lds_doubles(g->n_atoms, g->n_desc, g->n_train, chunk) * 8;
    StageArgs a{*g, chunk, *st, *sc, dt, stage, energy_partials};
    // big molecules: LDS allows one workgroup per CU anyway, so give a geometry eight wavefronts instead of four
    if (big) {
        if (hipFuncSetAttribute((const void *)gdml_stage_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return sc_check_launch("sc_gdml_stage (LDS attribute)");
        hipLaunchKernelGGL(gdml_stage_kernel<512>, dim3(sc_dense_grid(st->n)), dim3(512), lds, (hipStream_t)stream, a);
    } else {
        if (hipFuncSetAttribute((const void *)gdml_stage_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return sc_check_launch("sc_gdml_stage (LDS attribute)");
        hipLaunchKernelGGL(gdml_stage_kernel<256>, dim3(sc_dense_grid(st->n)), dim3(256), lds, (hipStream_t)stream, a);
    }
    return sc_check_launch("sc_gdml_stage");
}
