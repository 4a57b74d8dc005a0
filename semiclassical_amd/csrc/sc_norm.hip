// Pairwise coherent-state sum  sum_ij wb_i wk_j exp( rs_i + rs_j + X1_i.Y1_j + i (ib_i + ik_j + X2_i.Y2_j) )
// -- the O(n^2) kernel behind HermanKlukPropagator.norm() (reference semiclassical/propagators.py:734-782).
//
// With A = Gt (2Gt)^+ Gt, B = (2Gt)^+, C = Gt (2Gt)^+ the overlap <q_i,p_i,Gt|q_j,p_j,Gt> (propagators.py:230-237)
// factorises into per-trajectory terms and five dot products per pair, which the host packs into two real
// "GEMM" operands:  X1 = [q, p], Y1 = [A q, B p] (real part) and X2 = [q, C p, q], Y2 = [p, -q, -C p] (imaginary part).
// One workgroup computes a 64 x 64 tile of pairs (4 x 4 pairs per thread, K staged through LDS in chunks of 16)
// and writes one complex partial sum; sc_reduce_slot adds the partials in fixed order.
#include "sc_common.h"

namespace {

struct PairArgs {
    const double *X1, *Y1, *X2, *Y2;    // [ni][K1], [nj][K1], [ni][K2], [nj][K2]
    int K1, K2;
    const double *rsi, *rsj, *ib, *ik;  // [ni], [nj], [ni], [nj] real
    const double *wb, *wk;              // [ni], [nj] complex
    int64_t ni, nj;                     // bras i (rows of the X operands), kets j (rows of the Y operands)
    double *partials;                   // [tiles][4], columns 2, 3 are zero
};

#define PT 64      // tile edge
#define PK 16      // K chunk

__global__ __launch_bounds__(256) void pair_sum_kernel(PairArgs A) {
    __shared__ double xs[PK][PT + 1], ys[PK][PT + 1];
    __shared__ double red[32];
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    const int64_t tiles = (A.nj + PT - 1) / PT;
    const int64_t bi = blockIdx.x / tiles, bj = blockIdx.x % tiles;
    const int64_t i0 = bi * PT, j0 = bj * PT;
    double re[4][4], im[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { re[a][b] = 0.0; im[a][b] = 0.0; }
    for (int part = 0; part < 2; ++part) {
        const double *X = part ? A.X2 : A.X1, *Y = part ? A.Y2 : A.Y1;
        const int K = part ? A.K2 : A.K1;
        for (int k0 = 0; k0 < K; k0 += PK) {
            __syncthreads();
            for (int e = tid; e < PK * PT; e += 256) {
                const int r = e / PK, k = e - r * PK;            // row r of the tile, column k of the chunk
                const bool kin = k0 + k < K;
                xs[k][r] = (kin && i0 + r < A.ni) ? X[(i0 + r) * K + k0 + k] : 0.0;
                ys[k][r] = (kin && j0 + r < A.nj) ? Y[(j0 + r) * K + k0 + k] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < PK; ++k) {
                double xv[4], yv[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) { xv[a] = xs[k][4 * ti + a]; yv[a] = ys[k][4 * tj + a]; }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        if (part) im[a][b] = fma(xv[a], yv[b], im[a][b]);
                        else re[a][b] = fma(xv[a], yv[b], re[a][b]);
                    }
            }
        }
    }
    double acc[2] = {0.0, 0.0};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int64_t i = i0 + 4 * ti + a;
        if (i >= A.ni) continue;
        const cplx wb = ((const cplx *)A.wb)[i];
        const double rsi = A.rsi[i], ibi = A.ib[i];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int64_t j = j0 + 4 * tj + b;
            if (j >= A.nj) continue;
            const cplx e = c_exp(c_make(rsi + A.rsj[j] + re[a][b], ibi + A.ik[j] + im[a][b]));
            const cplx t = c_mul(c_mul(wb, ((const cplx *)A.wk)[j]), e);
            acc[0] += t.x; acc[1] += t.y;
        }
    }
    block_sum<2>(acc, red);
    if (tid == 0) {
        A.partials[(size_t)blockIdx.x * 4 + 0] = acc[0];
        A.partials[(size_t)blockIdx.x * 4 + 1] = acc[1];
        A.partials[(size_t)blockIdx.x * 4 + 2] = 0.0;
        A.partials[(size_t)blockIdx.x * 4 + 3] = 0.0;
    }
}

// Frozen-Gaussian wavefunction on a spatial grid (reference propagators.py:252-292, 688-732)
//   phi(x_k) = fac sum_n v_n exp( -1/2 |L x_k - L q_n|^2 + i ( p_n . x_k - p_n . q_n ) ),   L = Gamma_t^(1/2) (symmetric),
// i.e. the quadratic form (x-q)^T Gamma_t (x-q) written as a squared distance in the L-transformed coordinates, which
// the host prepares once per call (LqT, PmT are [D][n], trajectory index fastest: coalesced; pq[n] = p_n . q_n).
// One workgroup owns GX grid points (their L x and x staged in LDS) and strides over the trajectories.
#define GX 4
struct GridArgs {
    const double *LqT, *PmT, *pq, *v;      // [D][n], [D][n], [n], [n] complex
    int64_t n;
    int D, nx;
    const double *Lx, *X;                 // [nx][D]
    double fac;
    double *phi;                          // [nx] complex
};

__global__ __launch_bounds__(256) void grid_sum_kernel(GridArgs A) {
    extern __shared__ double gs[];        // Lx[GX][D], X[GX][D]
    __shared__ double red[32];
    const int tid = threadIdx.x, D = A.D, k0 = blockIdx.x * GX;
    for (int e = tid; e < GX * D; e += 256) {
        const int g = e / D, a = e - g * D;
        const bool in = k0 + g < A.nx;
        gs[e] = in ? A.Lx[(size_t)(k0 + g) * D + a] : 0.0;
        gs[GX * D + e] = in ? A.X[(size_t)(k0 + g) * D + a] : 0.0;
    }
    __syncthreads();
    double acc[2 * GX];
#pragma unroll
    for (int g = 0; g < 2 * GX; ++g) acc[g] = 0.0;
    for (int64_t i = tid; i < A.n; i += 256) {
        double e2[GX], ph[GX];
        const double pq = A.pq[i];
#pragma unroll
        for (int g = 0; g < GX; ++g) { e2[g] = 0.0; ph[g] = -pq; }
        for (int a = 0; a < D; ++a) {
            const double lq = A.LqT[(size_t)a * A.n + i], p = A.PmT[(size_t)a * A.n + i];
#pragma unroll
            for (int g = 0; g < GX; ++g) {
                const double d = gs[g * D + a] - lq;
                e2[g] = fma(d, d, e2[g]);
                ph[g] = fma(p, gs[GX * D + g * D + a], ph[g]);
            }
        }
        const cplx v = ((const cplx *)A.v)[i];
#pragma unroll
        for (int g = 0; g < GX; ++g) {
            const cplx t = c_mul(v, c_exp(c_make(-0.5 * e2[g], ph[g])));
            acc[2 * g] += t.x; acc[2 * g + 1] += t.y;
        }
    }
    block_sum<2 * GX>(acc, red);
    if (tid == 0) {
#pragma unroll
        for (int g = 0; g < GX; ++g)
            if (k0 + g < A.nx) {
                A.phi[2 * (size_t)(k0 + g)] = A.fac * acc[2 * g];
                A.phi[2 * (size_t)(k0 + g) + 1] = A.fac * acc[2 * g + 1];
            }
    }
}

}  // namespace

extern "C" int sc_grid_sum(const double *LqT, const double *PmT, const double *pq, const double *v, int64_t n, int32_t D,
                           const double *Lx, const double *X, int32_t nx, double fac, double *phi, void *stream) {
    if (!LqT || !PmT || !pq || !v || !Lx || !X || !phi) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_grid_sum: null argument");
    if (D < 1 || D > 512) return sc_fail(SC_ERR_UNSUPPORTED, "sc_grid_sum: D=%d outside 1..512", D);
    if (nx <= 0 || n <= 0) return SC_OK;
    GridArgs a{LqT, PmT, pq, v, n, D, nx, Lx, X, fac, phi};
    hipLaunchKernelGGL(grid_sum_kernel, dim3((unsigned)((nx + GX - 1) / GX)), dim3(256), 2 * GX * D * sizeof(double),
                       (hipStream_t)stream, a);
    return sc_check_launch("sc_grid_sum");
}

extern "C" int64_t sc_pair_sum_tiles(int64_t n) {
    const int64_t t = (n + PT - 1) / PT;
    return t * t;
}
extern "C" int64_t sc_pair_sum_rect_tiles(int64_t ni, int64_t nj) { return ((ni + PT - 1) / PT) * ((nj + PT - 1) / PT); }

extern "C" int sc_pair_sum_rect(const double *X1, const double *Y1, int32_t K1, const double *X2, const double *Y2, int32_t K2,
                                const double *rs_i, const double *rs_j, const double *ib, const double *ik, const double *wb,
                                const double *wk, int64_t ni, int64_t nj, double *partials, void *stream) {
    if (!X1 || !Y1 || !X2 || !Y2 || !rs_i || !rs_j || !ib || !ik || !wb || !wk || !partials)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_pair_sum: null argument");
    if (ni <= 0 || nj <= 0) return SC_OK;
    const int64_t tiles = sc_pair_sum_rect_tiles(ni, nj);
    if (tiles > 0x7fffffff) return sc_fail(SC_ERR_UNSUPPORTED, "sc_pair_sum: %lld x %lld pairs need more than 2^31 tiles", (long long)ni, (long long)nj);
    PairArgs a{X1, Y1, X2, Y2, K1, K2, rs_i, rs_j, ib, ik, wb, wk, ni, nj, partials};
    hipLaunchKernelGGL(pair_sum_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, a);
    return sc_check_launch("sc_pair_sum");
}

extern "C" int sc_pair_sum(const double *X1, const double *Y1, int32_t K1, const double *X2, const double *Y2, int32_t K2,
                           const double *rs, const double *ib, const double *ik, const double *wb, const double *wk,
                           int64_t n, double *partials, void *stream) {
    return sc_pair_sum_rect(X1, Y1, K1, X2, Y2, K2, rs, rs, ib, ik, wb, wk, n, n, partials, stream);
}
