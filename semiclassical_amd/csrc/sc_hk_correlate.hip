// Coherent-state overlaps, non-adiabatic coupling factors and the per-step reduction into C_auto / k_ic.
//
// Lane = mode, grid-stride over trajectories (overlap / nac kernels: one per wavefront pass; correlate: 16).  Reference
// semantics reproduced:
//   <q,p,Gbra|qk,pk,Gket>                         semiclassical/propagators.py:181-240
//   C_qp = conj(vt) vi (signs c) exp(iS/hbar)     semiclassical/propagators.py:784-807
//   Monte-Carlo weight 1/(n probi (2 pi hbar)^D)  semiclassical/propagators.py:837
//   nacQ, nacq, k_ic                              semiclassical/propagators.py:886-909
#include "sc_common.h"
#include "sc_row16.h"

namespace {

struct OverlapArgs {
    sc_overlap_consts oc;
    const double *qp;
    int64_t n;
    double *out;
};

// exponent sums of one overlap; lanes stride over modes.  `dvec` is per-wave LDS scratch (2*D) for the dense case.
// returns (sA, sB, sP, sC) summed over the wave.
__device__ __forceinline__ void overlap_sums(const sc_overlap_consts &oc, const double *qp, double *dvec,
                                             double &sA, double &sB, double &sP, double &sC) {
    const int D = oc.dim, lane = threadIdx.x & 63;
    sA = sB = sP = sC = 0.0;
    if (oc.diag) {
        for (int a = lane; a < D; a += 64) {
            const double dq = oc.qk[a] - qp[a], dpp = oc.pk[a] - qp[D + a];
            sA = fma(dq * oc.A[a], dq, sA);
            sB = fma(dpp * oc.B[a], dpp, sB);
            sP = fma(oc.pk[a], dq, sP);
            sC = fma(dq * oc.C[a], dpp, sC);
        }
    } else {
        for (int a = lane; a < D; a += 64) {
            dvec[a] = oc.qk[a] - qp[a];
            dvec[D + a] = oc.pk[a] - qp[D + a];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the wave's own LDS writes have landed
        for (int a = lane; a < D; a += 64) {
            double ya = 0, yb = 0, yc = 0;
            for (int b = 0; b < D; ++b) {
                ya = fma(oc.A[a * D + b], dvec[b], ya);
                yb = fma(oc.B[a * D + b], dvec[D + b], yb);
                yc = fma(oc.C[a * D + b], dvec[D + b], yc);
            }
            const double dq = dvec[a], dpp = dvec[D + a];
            sA = fma(dq, ya, sA);
            sB = fma(dpp, yb, sB);
            sP = fma(oc.pk[a], dq, sP);
            sC = fma(dq, yc, sC);
        }
        __builtin_amdgcn_wave_barrier();
    }
    sA = wave_sum(sA); sB = wave_sum(sB); sP = wave_sum(sP); sC = wave_sum(sC);
}

__device__ __forceinline__ cplx overlap_value(const sc_overlap_consts &oc, double sA, double sB, double sP, double sC) {
    const cplx ex = c_make(-0.5 * sA - 0.5 / (SC_HBAR * SC_HBAR) * sB, (-sP + sC) / SC_HBAR);
    return c_scale(c_exp(ex), oc.fac);
}

__global__ __launch_bounds__(256) void overlap_kernel(OverlapArgs A) {
    extern __shared__ double smem[];
    const int D = A.oc.dim, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    double *dvec = smem + (size_t)wave * 2 * D;
    for (int64_t tr = (int64_t)blockIdx.x * nw + wave; tr < A.n; tr += (int64_t)gridDim.x * nw) {
        double sA, sB, sP, sC;
        overlap_sums(A.oc, A.qp + tr * 2 * D, dvec, sA, sB, sP, sC);
        if (lane == 0) ((cplx *)A.out)[tr] = overlap_value(A.oc, sA, sB, sP, sC);
    }
}

struct NacArgs {
    sc_nac_consts nc;
    const double *zi;
    int64_t n;
    double *nacq;
};

__global__ __launch_bounds__(256) void nac_initial_kernel(NacArgs A) {
    const int D = A.nc.dim, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    for (int64_t tr = (int64_t)blockIdx.x * nw + wave; tr < A.n; tr += (int64_t)gridDim.x * nw) {
        const double *z = A.zi + tr * 2 * D;
        double sR = 0, sG = 0;
        for (int a = lane; a < D; a += 64) {
            sR = fma(A.nc.q0[a] - z[a], A.nc.rn[a], sR);
            sG = fma(z[D + a] - A.nc.p0[a], A.nc.gn[a], sG);
        }
        sR = wave_sum(sR); sG = wave_sum(sG);
        if (lane == 0) ((cplx *)A.nacq)[tr] = c_make(A.nc.n2 + sR, (A.nc.p0n1 + sG) / SC_HBAR);
    }
}

struct CorrArgs {
    sc_state st;
    sc_overlap_consts oc;
    sc_nac_consts nc;
    int has_nac;
    const double *vi, *probi, *nacq;
    double mc_norm;
    double *cq_out, *kq_out, *partials;
};

// A wavefront takes 16 consecutive trajectories: their exponent sums are reduced one after the other (lane = mode) and
// parked in lanes 0..15, then those 16 lanes evaluate the scalar tails -- complex exp, sqrt, the phase, the weight --
// side by side.  (One trajectory per pass left 63 lanes idle during a tail that is longer than the reductions.)
__global__ __launch_bounds__(256) void hk_correlate_kernel(CorrArgs A) {
    extern __shared__ double smem[];
    __shared__ double wsum[4][4];
    constexpr int B = 16;
    const int D = A.st.dim, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    double *dvec = smem + (size_t)wave * 2 * D;
    double acc[4] = {0, 0, 0, 0};
    const int64_t n = A.st.n, ntask = (n + B - 1) / B;
    for (int64_t task = (int64_t)blockIdx.x * nw + wave; task < ntask; task += (int64_t)gridDim.x * nw) {
        const int64_t base = task * B;
        const int cnt = (int)(n - base < B ? n - base : B);
        double mA = 0, mB = 0, mP = 0, mC = 0, mR = 0, mG = 0;
        if (A.oc.diag && D <= 64) {
            // diagonal widths, one mode per lane: (q_a, p_a) of trajectory j + 1 are requested before the six wave sums of
            // trajectory j run (the sums are a dependent DPP chain; the load latency hides behind it)
            const bool own = lane < D;
            const double ocA = own ? A.oc.A[lane] : 0.0, ocB = own ? A.oc.B[lane] : 0.0, ocC = own ? A.oc.C[lane] : 0.0;
            const double qk = own ? A.oc.qk[lane] : 0.0, pk = own ? A.oc.pk[lane] : 0.0;
            const double nq0 = (own && A.has_nac) ? A.nc.q0[lane] : 0.0, np0 = (own && A.has_nac) ? A.nc.p0[lane] : 0.0;
            const double nrn = (own && A.has_nac) ? A.nc.rn[lane] : 0.0, ngn = (own && A.has_nac) ? A.nc.gn[lane] : 0.0;
            const double *qp0 = A.st.qp + base * 2 * D;
            double qn = own ? qp0[lane] : 0.0, pn = own ? qp0[D + lane] : 0.0;
            for (int j = 0; j < cnt; ++j) {
                const double q = qn, p = pn;
                if (j + 1 < cnt) {
                    const double *nx = A.st.qp + (base + j + 1) * 2 * D;
                    qn = own ? nx[lane] : 0.0; pn = own ? nx[D + lane] : 0.0;
                }
                const double dq = qk - q, dpp = pk - p;
                double sA = wave_sum(dq * ocA * dq), sB = wave_sum(dpp * ocB * dpp), sP = wave_sum(pk * dq), sC = wave_sum(dq * ocC * dpp);
                double sR = 0, sG = 0;
                if (A.has_nac) { sR = wave_sum((nq0 - q) * nrn); sG = wave_sum((p - np0) * ngn); }
                if (lane == j) { mA = sA; mB = sB; mP = sP; mC = sC; mR = sR; mG = sG; }
            }
        } else {
            for (int j = 0; j < cnt; ++j) {
                const double *qp = A.st.qp + (base + j) * 2 * D;
                double sA, sB, sP, sC, sR = 0, sG = 0;
                overlap_sums(A.oc, qp, dvec, sA, sB, sP, sC);
                if (A.has_nac) {
                    for (int a = lane; a < D; a += 64) {
                        sR = fma(A.nc.q0[a] - qp[a], A.nc.rn[a], sR);
                        sG = fma(qp[D + a] - A.nc.p0[a], A.nc.gn[a], sG);
                    }
                    sR = wave_sum(sR); sG = wave_sum(sG);
                }
                if (lane == j) { mA = sA; mB = sB; mP = sP; mC = sC; mR = sR; mG = sG; }
            }
        }
        if (lane < cnt) {
            const int64_t tr = base + lane;
            const cplx vt = overlap_value(A.oc, mA, mB, mP, mC);
            const cplx c = c_scale(c_sqrt(((const cplx *)A.st.c2)[tr]), A.st.sgn[tr]);
            const cplx ph = c_exp(c_make(0.0, A.st.act[tr] / SC_HBAR));
            const double w = 1.0 / (A.mc_norm * A.probi[tr]);
            cplx cq = c_mul(c_mul(c_conj(vt), ((const cplx *)A.vi)[tr]), c_mul(c, ph));
            cq = c_scale(cq, w);
            acc[0] += cq.x; acc[1] += cq.y;
            if (A.cq_out) ((cplx *)A.cq_out)[tr] = cq;
            if (A.has_nac) {
                const cplx nacQ = c_make(A.nc.n2 + mR, -(A.nc.p0n1 + mG) / SC_HBAR);
                cplx kq = c_mul(c_mul(nacQ, ((const cplx *)A.nacq)[tr]), cq);
                kq = c_scale(kq, 1.0 / (SC_HBAR * SC_HBAR));
                acc[2] += kq.x; acc[3] += kq.y;
                if (A.kq_out) ((cplx *)A.kq_out)[tr] = kq;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = wave_sum(acc[i]);      // fixed order: deterministic
    if (lane == 0) { for (int i = 0; i < 4; ++i) wsum[wave][i] = acc[i]; }
    __syncthreads();
    if (threadIdx.x < 4) {
        double s = 0;
        for (int w = 0; w < nw; ++w) s += wsum[w][threadIdx.x];
        A.partials[(size_t)blockIdx.x * 4 + threadIdx.x] = s;
    }
}

// Dense (or rank-deficient) width matrices, D <= 16 -- methylium with the reference's Cartesian widths: ONE trajectory per
// 16-lane DPP row.  Lane a keeps row a of the overlap's matrices A, B, C in registers; y_a = sum_b A_ab dq_b takes the
// displacement of lane b inside the multiply-add (v_fmac_f64_dpp row_newbcast, sc_row16.h), the six sums over the modes
// are fused broadcast multiply-adds with 1.0.  A wavefront takes 64 consecutive trajectories, row g the 16 from
// base + 16 g one after the other; lane j of a row parks the sums of the row's j-th trajectory, so that afterwards all 64
// lanes evaluate the scalar tails (complex exp, sqrt, phase, weight) side by side.  Replaces, for these shapes, the branch
// of hk_correlate_kernel that runs one trajectory per wavefront pass with 12 of 64 lanes and re-reads A, B, C from memory
// per trajectory (0.67 ms per step at n = 1e5, three times the step kernel).
__global__ __launch_bounds__(256) void hk_correlate_rows16_kernel(CorrArgs A) {
    __shared__ double wsum[4][4];
    const int D = A.st.dim, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = tid & 15, g = (tid >> 4) & 3;
    const bool own = r < D, nac = A.has_nac != 0;
    double rowA[16], rowB[16], rowC[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const bool in = own && b < D;
        rowA[b] = in ? A.oc.A[r * D + b] : 0.0; rowB[b] = in ? A.oc.B[r * D + b] : 0.0; rowC[b] = in ? A.oc.C[r * D + b] : 0.0;
    }
    const double qk = own ? A.oc.qk[r] : 0.0, pk = own ? A.oc.pk[r] : 0.0;
    const double nq0 = (own && nac) ? A.nc.q0[r] : 0.0, np0 = (own && nac) ? A.nc.p0[r] : 0.0;
    const double nrn = (own && nac) ? A.nc.rn[r] : 0.0, ngn = (own && nac) ? A.nc.gn[r] : 0.0;
    double one = 1.0;
    asm volatile("" : "+v"(one));
    double acc[4] = {0, 0, 0, 0};
    const int64_t n = A.st.n, nbatch = (n + 63) / 64;
    for (int64_t batch = (int64_t)blockIdx.x * 4 + wave; batch < nbatch; batch += (int64_t)gridDim.x * 4) {
        const int64_t base = batch * 64 + 16 * g;
        double m[6] = {0, 0, 0, 0, 0, 0};
        auto fetch = [&](int j, double &q, double &p) {
            const int64_t t = base + j;
            const bool ok = own && t < n;
            q = ok ? A.st.qp[t * 2 * D + r] : qk;
            p = ok ? A.st.qp[t * 2 * D + D + r] : pk;
        };
        double qn, pn;
        fetch(0, qn, pn);
        for (int j = 0; j < 16; ++j) {
            const double q = qn, p = pn;
            if (j + 1 < 16) fetch(j + 1, qn, pn);              // requested before the dependent chains of trajectory j
            double dq = qk - q, dpp = pk - p;
            double ya = 0.0, yb = 0.0, yc = 0.0;
            dpp_guard(dq, dpp);
            sfor<0, 16>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                if (b < D) { fmac_bc<b>(ya, dq, rowA[b]); fmac_bc<b>(yb, dpp, rowB[b]); fmac_bc<b>(yc, dpp, rowC[b]); }
            });
            double t[6] = {dq * ya, dpp * yb, pk * dq, dq * yc, (nq0 - q) * nrn, (p - np0) * ngn}, s[6] = {0, 0, 0, 0, 0, 0};
            dpp_guard(t);
            sfor<0, 16>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if (k < D) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) fmac_bc<k>(s[i], t[i], one);
                }
            });
            if (r == j) {
#pragma unroll
                for (int i = 0; i < 6; ++i) m[i] = s[i];
            }
        }
        const int64_t tr = batch * 64 + lane;                  // lane (g, r) parked trajectory base + r
        if (tr < n) {
            const cplx vt = overlap_value(A.oc, m[0], m[1], m[2], m[3]);
            const cplx c = c_scale(c_sqrt(((const cplx *)A.st.c2)[tr]), A.st.sgn[tr]);
            const cplx ph = c_exp(c_make(0.0, A.st.act[tr] / SC_HBAR));
            const double w = 1.0 / (A.mc_norm * A.probi[tr]);
            cplx cq = c_mul(c_mul(c_conj(vt), ((const cplx *)A.vi)[tr]), c_mul(c, ph));
            cq = c_scale(cq, w);
            acc[0] += cq.x; acc[1] += cq.y;
            if (A.cq_out) ((cplx *)A.cq_out)[tr] = cq;
            if (nac) {
                const cplx nacQ = c_make(A.nc.n2 + m[4], -(A.nc.p0n1 + m[5]) / SC_HBAR);
                cplx kq = c_mul(c_mul(nacQ, ((const cplx *)A.nacq)[tr]), cq);
                kq = c_scale(kq, 1.0 / (SC_HBAR * SC_HBAR));
                acc[2] += kq.x; acc[3] += kq.y;
                if (A.kq_out) ((cplx *)A.kq_out)[tr] = kq;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = wave_sum(acc[i]);      // fixed order: deterministic
    if (lane == 0) { for (int i = 0; i < 4; ++i) wsum[wave][i] = acc[i]; }
    __syncthreads();
    if (threadIdx.x < 4) {
        double sum = 0;
        for (int w = 0; w < 4; ++w) sum += wsum[w][threadIdx.x];
        A.partials[(size_t)blockIdx.x * 4 + threadIdx.x] = sum;
    }
}

struct ReduceArgs {
    const double *cpart;
    int ncorr;
    const double *epart;
    int nen;
    double n_energy;
    double *slot;
    long long *cursor;      // optional device-resident row counter: the sums go to slot + 5 * (*cursor), then ++(*cursor)
};

__global__ __launch_bounds__(256) void reduce_slot_kernel(ReduceArgs A) {
    __shared__ double red[32];
    double v[5] = {0, 0, 0, 0, 0};
    if (A.cpart)
        for (int i = threadIdx.x; i < A.ncorr; i += blockDim.x)
            for (int k = 0; k < 4; ++k) v[k] += A.cpart[(size_t)i * 4 + k];
    if (A.epart)
        for (int i = threadIdx.x; i < A.nen; i += blockDim.x) v[4] += A.epart[i];
    block_sum<5>(v, red);
    if (threadIdx.x == 0) {
        double *slot = A.slot;
        if (A.cursor) { slot += 5 * *A.cursor; *A.cursor += 1; }
        if (A.cpart) for (int k = 0; k < 4; ++k) slot[k] = v[k];
        if (A.epart) slot[4] = v[4] / A.n_energy;
    }
}

__global__ __launch_bounds__(256) void energy_guard_kernel(const double *epart, int nblk, double n, double *elog) {
    __shared__ double red[8];
    double v[1] = {0.0};
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) v[0] += epart[i];
    block_sum<1>(v, red);
    if (threadIdx.x == 0) {
        const double mean = v[0] / n, prev = elog[1], count = elog[3];
        elog[0] = prev;
        elog[1] = mean;
        if (count >= 1.0) {
            const double change = fabs(mean - prev);
            if (change > elog[2]) elog[2] = change;
        }
        elog[3] = count + 1.0;
    }
}

// (rows, n) with n fastest  <->  (n, rows) with rows fastest, tiled through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const double *src, double *dst, int64_t rows, int64_t cols,
                                                        int64_t src_ld, int64_t dst_ld) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    for (int j = ty; j < 32; j += 8) {
        const int64_t r = r0 + j, c = c0 + tx;
        if (r < rows && c < cols) tile[j][tx] = src[r * src_ld + c];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int64_t c = c0 + j, r = r0 + tx;
        if (r < rows && c < cols) dst[c * dst_ld + r] = tile[tx][j];
    }
}

int launch_transpose(const double *src, double *dst, int64_t rows, int64_t cols, int64_t src_ld, int64_t dst_ld,
                     hipStream_t s) {
    if (rows <= 0 || cols <= 0) return SC_OK;
    int64_t by = (rows + 31) / 32, bx = (cols + 31) / 32;
    // grid.y is limited to 65535: split the row range
    for (int64_t yb = 0; yb < by; yb += 65535) {
        int64_t ny = by - yb < 65535 ? by - yb : 65535;
        int64_t r_off = yb * 32;
        int64_t nrows = rows - r_off < ny * 32 ? rows - r_off : ny * 32;
        hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)bx, (unsigned)ny), dim3(256), 0, s,
                           src + r_off * src_ld, dst + r_off, nrows, cols, src_ld, dst_ld);
    }
    return sc_check_launch("transpose");
}

}  // namespace

extern "C" int sc_correlate_grid(int64_t n, int32_t dim) {
    (void)dim;
    int64_t blocks = (n + 3) / 4;
    if (blocks < 1) blocks = 1;
    return (int)(blocks < 2048 ? blocks : 2048);
}

static size_t wave_scratch_bytes(int D, int diag) { return diag ? 0 : (size_t)4 * 2 * D * sizeof(double); }

extern "C" int sc_overlap(const sc_overlap_consts *oc, const double *qp, int64_t n, double *out, void *stream) {
    if (!oc || !qp || !out) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_overlap: null argument");
    if (n <= 0) return SC_OK;
    OverlapArgs a{*oc, qp, n, out};
    hipLaunchKernelGGL(overlap_kernel, dim3(sc_correlate_grid(n, oc->dim)), dim3(256),
                       wave_scratch_bytes(oc->dim, oc->diag), (hipStream_t)stream, a);
    return sc_check_launch("sc_overlap");
}

extern "C" int sc_nac_initial(const sc_nac_consts *nc, const double *zi, int64_t n, double *nacq, void *stream) {
    if (!nc || !zi || !nacq) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_nac_initial: null argument");
    if (n <= 0) return SC_OK;
    NacArgs a{*nc, zi, n, nacq};
    hipLaunchKernelGGL(nac_initial_kernel, dim3(sc_correlate_grid(n, nc->dim)), dim3(256), 0, (hipStream_t)stream, a);
    return sc_check_launch("sc_nac_initial");
}

extern "C" int sc_hk_correlate(const sc_state *st, const sc_overlap_consts *ovl_t0, const sc_nac_consts *nc,
                               const double *vi, const double *probi, const double *nacq, double mc_norm,
                               double *cq_out, double *kq_out, double *partials, void *stream) {
    if (!st || !ovl_t0 || !vi || !probi || !partials)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_correlate: null argument");
    if (nc && !nacq) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_correlate: nac constants without nacq");
    if (ovl_t0->dim != st->dim) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_correlate: dimension mismatch");
    CorrArgs a;
    a.st = *st; a.oc = *ovl_t0; a.has_nac = nc != nullptr;
    if (nc) a.nc = *nc; else a.nc = sc_nac_consts{};
    a.vi = vi; a.probi = probi; a.nacq = nacq; a.mc_norm = mc_norm;
    a.cq_out = cq_out; a.kq_out = kq_out; a.partials = partials;
    if (!ovl_t0->diag && st->dim <= 16)
        hipLaunchKernelGGL(hk_correlate_rows16_kernel, dim3(sc_correlate_grid(st->n, st->dim)), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(hk_correlate_kernel, dim3(sc_correlate_grid(st->n, st->dim)), dim3(256),
                           wave_scratch_bytes(st->dim, ovl_t0->diag), (hipStream_t)stream, a);
    return sc_check_launch("sc_hk_correlate");
}

extern "C" int sc_reduce_slot(const double *corr_partials, int32_t n_corr, const double *energy_partials,
                              int32_t n_energy_blocks, double n_energy, double *slot, void *stream) {
    if (!slot) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_reduce_slot: null slot");
    ReduceArgs a{corr_partials, n_corr, energy_partials, n_energy_blocks, n_energy, slot, nullptr};
    hipLaunchKernelGGL(reduce_slot_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    return sc_check_launch("sc_reduce_slot");
}

extern "C" int sc_reduce_slot_at(const double *corr_partials, int32_t n_corr, double *slots, int64_t *cursor, void *stream) {
    if (!slots || !cursor || !corr_partials) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_reduce_slot_at: null argument");
    ReduceArgs a{corr_partials, n_corr, nullptr, 0, 1.0, slots, (long long *)cursor};
    hipLaunchKernelGGL(reduce_slot_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
    return sc_check_launch("sc_reduce_slot_at");
}

extern "C" int sc_energy_guard(const double *energy_partials, int32_t n_blocks, double n_traj, double *elog,
                               void *stream) {
    if (!energy_partials || !elog) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_energy_guard: null argument");
    hipLaunchKernelGGL(energy_guard_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, energy_partials, n_blocks,
                       n_traj, elog);
    return sc_check_launch("sc_energy_guard");
}

extern "C" int sc_state_from_reference(const double *y, const sc_state *st, void *stream) {
    if (!y || !st) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_state_from_reference: null argument");
    if (int rq = sc_require_rowmajor(st, "sc_state_from_reference")) return rq;
    const int64_t D = st->dim, n = st->n, DD = D * D;
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_transpose(y, st->qp, 2 * D, n, n, 2 * D, s);
    if (rc) return rc;
    rc = launch_transpose(y + 2 * D * n, st->mono, 4 * DD, n, n, 4 * DD, s);
    if (rc) return rc;
    return launch_transpose(y + (2 * D + 4 * DD) * n, st->act, 1, n, n, 1, s);
}

extern "C" int sc_state_to_reference(const sc_state *st, double *y, void *stream) {
    if (!y || !st) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_state_to_reference: null argument");
    if (int rq = sc_require_rowmajor(st, "sc_state_to_reference")) return rq;
    const int64_t D = st->dim, n = st->n, DD = D * D;
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_transpose(st->qp, y, n, 2 * D, 2 * D, n, s);
    if (rc) return rc;
    rc = launch_transpose(st->mono, y + 2 * D * n, n, 4 * DD, 4 * DD, n, s);
    if (rc) return rc;
    return launch_transpose(st->act, y + (2 * D + 4 * DD) * n, n, 1, 1, n, s);
}
