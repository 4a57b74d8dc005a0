// Walton-Manolopoulos prefactor and per-trajectory correlation terms, fused per trajectory.
//
// One workgroup per trajectory (grid-stride); every matrix of the trajectory lives in LDS.  Called after the HK step
// kernel, which has already advanced (q, p, S, M) and tracked sqrt(c2).  Reference semantics reproduced
// (paths relative to the reference repository, equation numbers of Walton & Manolopoulos 1996 as cited there):
//   _expand_L        semiclassical/propagators.py:1132-1193   grad / Hessian of i/hbar S
//   _prefactor       semiclassical/propagators.py:1195-1389   A (50), BQ Bq b0 (53-55), Gt Gti (57-59), CQQ CqQ PIq PIQ eps
//                                                             (69-74), M (78), R.. P. gamma (79-84), trackers of detA, detM
//   autocorrelation_qp  :1577-1614  eqn (85)      ic_correlation  :1652-1719  eqn (100)
//
// Restatement used here (all contractions are exact rewrites; differences are re-association only):
//   * everything is carried in the projected space e = 2d':  Mq' = [Mqq U, Mqp U], Mp' = [Mpq U, Mpp U]  (D x e, real)
//       A' = Cst' + Mq'^T Gt_w Mq' + i/hbar (2G - H),   G = Mp'^T Mq',  H[i][j] = i < d' ? G[i][j] : G[j][i]
//       Cst' = 2 blockdiag(alpha U^T G0 U, beta U^T G0^+ U) + blockdiag(U^T Gi U, 0) - 2i/hbar [[0,0],[1,0]]      (host)
//     the reference's b0 (55) is identically zero analytically (its two terms are the same contraction), so
//     pi_t = P and pi_i = p;
//   * A'/(2 sqrt(alpha beta)) and M'/(2 pi) are inverted by Gauss-Jordan with partial pivoting, which also yields the
//     determinants detA, detM the reference takes of the same scaled matrices (:1328-1332, 1358-1359);
//   * Rqq, RQQ, RqQ, Pq, PQ are never formed: eqn (85)/(100) only need their contractions with dq = q0-q, dQ = q0-Q and
//     n1, i.e. bilinear forms a^T iM b with a, b in { CqQ^T dq, CqQ^T n1, G0 dQ, G0 n1, PIQ - p0 }.
//
// Three mappings of this arithmetic exist (sc_wm_correlate picks one):
//   wm_small_kernel<D, d'>  (sc_wm_small.hip)  D <= 16, e <= 16 at the instantiated shapes: registers + DPP
//   wm_kernel<false>        every matrix of the trajectory in LDS (one workgroup per trajectory)
//   wm_kernel<true>         the same code with the matrices in a per-workgroup block of GLOBAL memory (L2 / MALL
//                           resident scratch supplied by the caller) for shapes whose matrices exceed the 160 KB of LDS
#include "sc_wm.h"

namespace {

// Gauss-Jordan inversion of the n x n complex matrix in the left half of aug (n x 2n, row-major); the right half must
// hold the identity on entry and holds the inverse on exit.  Returns det(left) to every thread.  colbuf: n complex.
__device__ __forceinline__ cplx lds_gauss_jordan(cplx *aug, int n, cplx *colbuf, int *ipiv) {
    const int tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, w2 = 2 * n;
    cplx det = c_make(1.0, 0.0);
    for (int k = 0; k < n; ++k) {
        if (tid < 64) {
            // one candidate row per lane (n <= 64): |a_ik|^2 of row i = k + lane
            const int i = k + lane;
            const bool valid = i < n;
            int bi = wave_pivot_row(valid ? c_abs2(aug[i * w2 + k]) : 0.0, i, valid);
            if (bi < 0) bi = k;
            if (bi != k) {
                for (int j = lane; j < w2; j += 64) {
                    const cplx t = aug[k * w2 + j];
                    aug[k * w2 + j] = aug[bi * w2 + j];
                    aug[bi * w2 + j] = t;
                }
            }
            if (lane == 0) *ipiv = bi;
        }
        __syncthreads();
        const cplx piv = aug[k * w2 + k];
        det = c_mul(det, piv);
        if (*ipiv != k) det = c_make(-det.x, -det.y);
        if (piv.x == 0.0 && piv.y == 0.0) { __syncthreads(); return c_make(0.0, 0.0); }
        const cplx inv = c_inv(piv);
        for (int i = tid; i < n; i += nth) colbuf[i] = aug[i * w2 + k];
        __syncthreads();
        for (int j = tid; j < w2; j += nth) aug[k * w2 + j] = c_mul(aug[k * w2 + j], inv);
        __syncthreads();
        {   // rows strided over groups of `wp` threads (wp = w2 rounded up to a power of two): no integer division
            int wp = 1;
            while (wp < w2) wp <<= 1;
            const int j = tid & (wp - 1), rstep = nth > wp ? nth / wp : 1, r0 = nth > wp ? tid / wp : 0;
            if (j < w2 && (nth >= wp || tid < wp)) {
                const cplx rk = aug[k * w2 + j];
                for (int i = r0; i < n; i += rstep)
                    if (i != k) aug[i * w2 + j] = c_fnma(colbuf[i], rk, aug[i * w2 + j]);
            }
        }
        __syncthreads();
    }
    return det;
}

// sizes of the shared LDS regions of wm_kernel (doubles / complex values), see the carve-up there
__host__ __device__ inline size_t wm_region_p(int D, int dp) {
    const size_t E = 2 * (size_t)dp, a = 3 * (size_t)D * E + E * E, b = 2 * (size_t)D * D + 4 * (size_t)dp * dp + 20 * (size_t)dp;
    const size_t m = a > b ? a : b;
    return m + (m & 1);
}
__host__ __device__ inline size_t wm_region_q(int D, int dp) {
    const size_t E = 2 * (size_t)dp, a = 2 * E * E, b = 2 * (size_t)D * D;
    return a > b ? a : b;
}

template <bool GLOBAL_SCRATCH>
__global__ __launch_bounds__(256) void wm_kernel(WmArgs A) {
    extern __shared__ double2 smem2[];
    __shared__ double red[32];
    __shared__ int ipiv;
    const int D = A.st.dim, dp = A.wc.dprime, E = 2 * dp, DD = D * D, tid = threadIdx.x, nth = blockDim.x;
    const sc_wm_consts &W = A.wc;

    // ---- LDS carve-up.  Arrays whose lifetimes do not overlap share storage (a trajectory then needs ~14 KB instead of
    // ~26 KB at D = 12, d' = 6, which is what limits the number of resident trajectories per CU):
    //   region P: {Mq', Mp', Gamma_t Mq', Mp'^T Mq'} (dead once A' and BQ' exist) -> Wm = BQ' iA' (dead once Gt, Gti
    //             exist) -> {V, [M'|I], hat, rho}
    //   region Q: [A'/s | I] (dead once Wm exists) -> {Gt (later CQQ), Gti}
    // GLOBAL_SCRATCH: the same carve-up inside this workgroup's block of the caller's scratch buffer.  The waves of a
    // workgroup share one CU and its write-through L1, so __syncthreads() orders these global accesses as it orders LDS.
    double *f = GLOBAL_SCRATCH ? (double *)((char *)A.scratch + (size_t)blockIdx.x * A.scratch_stride) : (double *)smem2;
    double *vec = f;             f += 8 * D;        // dq, dQ, dp, g, s_dq, w_dQ, (2 spare)
    double *rowtmp = f;          f += 3 * D + (D & 1);   // per-row partial results of the scalar tail (3 x D)
    double *P = f;               f += wm_region_p(D, dp);
    double *Mq = P, *Mp = Mq + D * E, *Tq = Mp + D * E, *G = Tq + D * E;
    cplx *Wm = (cplx *)P;                             // BQ' iA'
    cplx *V = (cplx *)P, *augM = V + DD, *hat = augM + dp * 2 * dp, *rho = hat + 5 * dp;
    cplx *c = (cplx *)f;
    cplx *aug = c;               c += wm_region_q(D, dp);   // [A'/s | I] -> [I | s iA']
    cplx *Gt = aug, *Gti = aug + DD;                  // Gt, later CQQ
    cplx *BQ = c;                c += D * E;
    cplx *colbuf = c;            c += E > D ? E : D;
    cplx *cv = c;                c += 5 * D;        // u_dq, u_n1, w_dQ(c), w_n1(c), y
    double *cst = (double *)c;                       // optional LDS copies of the constants

    // constants: global pointers, or LDS copies when the host found room for them
    const double *cU = W.U, *cGt = W.Gt, *cG0 = W.G0, *ciGi0 = W.iGi0, *cS = W.S, *cCqq = W.Cqq;
    const double *cq0 = W.q0, *cp0 = W.p0, *cn1 = W.n1, *csn1 = W.s_n1, *cwn1 = W.w_n1;
    // re-run of the trajectories the register kernel flagged (fixed pivot order too weak): nothing to do in the common case,
    // not even the staging of the constants
    const int32_t *only = A.only_flagged;
    const bool idle = only && only[A.st.n] == 0;
    if (A.stage_consts && !idle) {
        double *d = cst;
        auto stage = [&](const double *&ptr, int count) {
            if (ptr) { for (int i = tid; i < count; i += nth) d[i] = ptr[i]; ptr = d; }
            d += count;
        };
        stage(cU, D * dp); stage(cGt, DD); stage(cG0, DD); stage(ciGi0, DD); stage(cS, DD); stage(cCqq, DD);
        stage(cq0, D); stage(cp0, D); stage(cn1, D); stage(csn1, D); stage(cwn1, D);
        __syncthreads();
    }

    const cplx *Cst = (const cplx *)W.Cst, *Bq = (const cplx *)W.Bq;
    const double ihb = 1.0 / SC_HBAR;
    double acc[4] = {0, 0, 0, 0};

    for (int64_t tr = blockIdx.x; tr < A.st.n && !idle; tr += gridDim.x) {
        // coupling vectors: the constants of the reference potentials, or (position-dependent derivative couplings,
        // sc_wm_consts.nac_traj) this trajectory's own n1(q_i), S n1(q_i), G0 n1(Q), p0.n1(Q), n2(q_i), n2(Q)
        const double *tn1 = cn1, *tsn1 = csn1, *twn1 = cwn1;
        double tp0n1 = W.p0n1, tn2q = W.n2, tn2Q = W.n2;
        if (W.nac_traj) {
            const double *blk = W.nac_traj + (size_t)tr * (3 * D + 3);
            tn1 = blk; tsn1 = blk + D; twn1 = blk + 2 * D;
            tp0n1 = blk[3 * D]; tn2q = blk[3 * D + 1]; tn2Q = blk[3 * D + 2];
        }
        if (only && !only[tr]) continue;
        const double *qp = A.st.qp + tr * 2 * D, *zi = A.zi + tr * 2 * D;
        const double *M = A.st.mono + tr * 4 * (int64_t)DD;
        __syncthreads();
        // Mq' = [Mqq U, Mqp U], Mp' = [Mpq U, Mpp U]; per-trajectory vectors
        for (int e = tid; e < D * E; e += nth) {
            const int a = e / E, j = e - a * E, blk = j >= dp, jj = blk ? j - dp : j;
            const double *Mqx = M + (blk ? DD : 0) + a * D, *Mpx = M + (blk ? 3 * DD : 2 * DD) + a * D;
            double sq = 0.0, sp = 0.0;
            for (int b = 0; b < D; ++b) { const double u = cU[b * dp + jj]; sq = fma(Mqx[b], u, sq); sp = fma(Mpx[b], u, sp); }
            Mq[e] = sq; Mp[e] = sp;
        }
        for (int a = tid; a < D; a += nth) {
            vec[a] = cq0[a] - zi[a];                 // dq
            vec[D + a] = cq0[a] - qp[a];             // dQ
            vec[2 * D + a] = cp0[a] - zi[D + a];     // dp = p0 - p_initial
        }
        __syncthreads();
        // Tq = Gamma_t Mq' ; G = Mp'^T Mq' ; g = iGi0 dp ; s_dq = S dq ; w_dQ = G0 dQ
        for (int e = tid; e < D * E; e += nth) {
            const int a = e / E, j = e - a * E;
            double s = 0.0;
            for (int b = 0; b < D; ++b) s = fma(cGt[a * D + b], Mq[b * E + j], s);
            Tq[e] = s;
        }
        for (int e = tid; e < E * E; e += nth) {
            const int i = e / E, j = e - i * E;
            double s = 0.0;
            for (int a = 0; a < D; ++a) s = fma(Mp[a * E + i], Mq[a * E + j], s);
            G[e] = s;
        }
        for (int a = tid; a < D; a += nth) {
            double g = 0.0, s = 0.0, w = 0.0;
            for (int b = 0; b < D; ++b) {
                g = fma(ciGi0[a * D + b], vec[2 * D + b], g);
                s = fma(cS[a * D + b], vec[b], s);
                w = fma(cG0[a * D + b], vec[D + b], w);
            }
            vec[3 * D + a] = g; vec[4 * D + a] = s; vec[5 * D + a] = w;
        }
        __syncthreads();
        // A'/s with s = 2 sqrt(alpha beta), identity on the right; BQ' = Tq + i/hbar Mp'
        for (int e = tid; e < E * E; e += nth) {
            const int i = e / E, j = e - i * E;
            double s = 0.0;
            for (int a = 0; a < D; ++a) s = fma(Mq[a * E + i], Tq[a * E + j], s);
            const double h = (i < dp) ? G[i * E + j] : G[j * E + i];
            const cplx a0 = Cst[e];
            aug[i * 2 * E + j] = c_make((a0.x + s) * W.inv_scale_a, (a0.y + ihb * (2.0 * G[e] - h)) * W.inv_scale_a);
            aug[i * 2 * E + E + j] = c_make(i == j ? 1.0 : 0.0, 0.0);
        }
        for (int e = tid; e < D * E; e += nth) BQ[e] = c_make(Tq[e], ihb * Mp[e]);
        __syncthreads();
        const cplx detA = lds_gauss_jordan(aug, E, colbuf, &ipiv);
        // Wm = BQ' iA' (iA' = inverse(A'/s)/s)
        for (int e = tid; e < D * E; e += nth) {
            const int a = e / E, j = e - a * E;
            cplx s = c_make(0, 0);
            for (int k = 0; k < E; ++k) s = c_fma(BQ[a * E + k], aug[k * 2 * E + E + j], s);
            Wm[e] = c_scale(s, W.inv_scale_a);
        }
        __syncthreads();
        // Gt = Gamma_t - Wm BQ'^T (57) ; Gti = Wm Bq'^T (59)
        for (int e = tid; e < DD; e += nth) {
            const int a = e / D, b = e - a * D;
            cplx s = c_make(0, 0), t = c_make(0, 0);
            for (int j = 0; j < E; ++j) {
                s = c_fma(Wm[a * E + j], BQ[b * E + j], s);
                t = c_fma(Wm[a * E + j], Bq[b * E + j], t);
            }
            Gt[e] = c_make(cGt[e] - s.x, -s.y);
            Gti[e] = t;
        }
        __syncthreads();
        // V = Gti iGi0 ; y = (P - p0) + Gti g ; u_dq = Gti s_dq ; u_n1 = Gti s_n1 ; w vectors as complex
        for (int e = tid; e < DD; e += nth) {
            const int a = e / D, b = e - a * D;
            cplx s = c_make(0, 0);
            for (int k = 0; k < D; ++k) { const double x = ciGi0[k * D + b]; s.x = fma(Gti[a * D + k].x, x, s.x); s.y = fma(Gti[a * D + k].y, x, s.y); }
            V[e] = s;
        }
        for (int a = tid; a < D; a += nth) {
            cplx y = c_make(qp[D + a] - cp0[a], 0.0), u1 = c_make(0, 0), u2 = c_make(0, 0);
            for (int b = 0; b < D; ++b) {
                const cplx gt = Gti[a * D + b];
                const double g = vec[3 * D + b], s1 = vec[4 * D + b], s2 = A.has_nac ? tsn1[b] : 0.0;
                y.x = fma(gt.x, g, y.x); y.y = fma(gt.y, g, y.y);
                u1.x = fma(gt.x, s1, u1.x); u1.y = fma(gt.y, s1, u1.y);
                u2.x = fma(gt.x, s2, u2.x); u2.y = fma(gt.y, s2, u2.y);
            }
            cv[a] = u1; cv[D + a] = u2;
            cv[2 * D + a] = c_make(vec[5 * D + a], 0.0);
            cv[3 * D + a] = c_make(A.has_nac ? twn1[a] : 0.0, 0.0);
            cv[4 * D + a] = y;
        }
        __syncthreads();
        // CQQ = Gt - V Gti^T (70), in place in Gt
        for (int e = tid; e < DD; e += nth) {
            const int a = e / D, b = e - a * D;
            cplx s = c_make(0, 0);
            for (int k = 0; k < D; ++k) s = c_fma(V[a * D + k], Gti[b * D + k], s);
            Gt[e] = c_sub(Gt[e], s);         // each thread only touches its own element of Gt
        }
        __syncthreads();
        if (W.cqq_out) {
            cplx *out = (cplx *)W.cqq_out + tr * (int64_t)DD;
            for (int e = tid; e < DD; e += nth) out[e] = Gt[e];
        }
        if (W.dvec_out) {                        // C_qQ^T (q0 - q) + i/hbar PI_Q = u_dq + i/hbar (y + p0)
            cplx *out = (cplx *)W.dvec_out + tr * (int64_t)D;
            for (int a = tid; a < D; a += nth)
                out[a] = c_make(cv[a].x - ihb * cv[4 * D + a].y, cv[a].y + ihb * (cv[4 * D + a].x + cp0[a]));
        }
        // M'/(2 pi) = U^T (G0 + CQQ) U / (2 pi) ; identity on the right ; hat = U^T {5 vectors}
        for (int e = tid; e < dp * dp; e += nth) {
            const int i = e / dp, j = e - i * dp;
            cplx s = c_make(0, 0);
            for (int a = 0; a < D; ++a) {
                cplx row = c_make(0, 0);
                for (int b = 0; b < D; ++b) {
                    const double u = cU[b * dp + j];
                    row.x = fma(cG0[a * D + b] + Gt[a * D + b].x, u, row.x);
                    row.y = fma(Gt[a * D + b].y, u, row.y);
                }
                const double ui = cU[a * dp + i];
                s.x = fma(ui, row.x, s.x); s.y = fma(ui, row.y, s.y);
            }
            augM[i * 2 * dp + j] = c_scale(s, W.inv_two_pi);
            augM[i * 2 * dp + dp + j] = c_make(i == j ? 1.0 : 0.0, 0.0);
        }
        for (int e = tid; e < 5 * dp; e += nth) {
            const int v = e / dp, i = e - v * dp;
            cplx s = c_make(0, 0);
            for (int a = 0; a < D; ++a) { const double u = cU[a * dp + i]; s.x = fma(u, cv[v * D + a].x, s.x); s.y = fma(u, cv[v * D + a].y, s.y); }
            hat[e] = s;
        }
        __syncthreads();
        const cplx detM = lds_gauss_jordan(augM, dp, colbuf, &ipiv);
        // rho_v = iM' hat_v,  iM' = inverse(M'/(2 pi)) / (2 pi)
        for (int e = tid; e < 5 * dp; e += nth) {
            const int v = e / dp, i = e - v * dp;
            cplx s = c_make(0, 0);
            for (int k = 0; k < dp; ++k) s = c_fma(augM[i * 2 * dp + dp + k], hat[v * dp + k], s);
            rho[e] = c_scale(s, W.inv_two_pi);
        }
        __syncthreads();

        // row sums of the D x D constant forms, one row per thread
        for (int a = tid; a < D; a += nth) {
            double cdq = 0.0, cn1v = 0.0, g0g = 0.0;
            for (int b = 0; b < D; ++b) {
                cdq = fma(cCqq[a * D + b], vec[b], cdq);
                if (A.has_nac) cn1v = fma(cCqq[a * D + b], tn1[b], cn1v);
                g0g = fma(cG0[a * D + b], vec[3 * D + b], g0g);
            }
            rowtmp[a] = cdq; rowtmp[D + a] = cn1v; rowtmp[2 * D + a] = g0g;
        }
        __syncthreads();

        // ---- scalars: one thread per trajectory is enough (O(d'^2 + D) work) ----
        if (tid == 0) {
            auto form = [&](int a, int b) {                       // a^T iM b
                cplx s = c_make(0, 0);
                for (int i = 0; i < dp; ++i) s = c_fma(hat[a * dp + i], rho[b * dp + i], s);
                return s;
            };
            enum { UDQ = 0, UN1 = 1, WDQ = 2, WN1 = 3, Y = 4 };
            double dqCdq = 0, dqCn1 = 0, dQGdQ = 0, dQGn1 = 0, piq_dq = 0, piq_n1 = 0, p0_dQ = 0, eps = 0;
            for (int a = 0; a < D; ++a) {
                const double cdq = rowtmp[a], crow = rowtmp[D + a], g0g = rowtmp[2 * D + a];
                const double piq = cp0[a] - g0g;                                  // (72)
                dqCdq = fma(vec[a], cdq, dqCdq);
                dqCn1 = fma(vec[a], crow, dqCn1);
                dQGdQ = fma(vec[D + a], vec[5 * D + a], dQGdQ);
                if (A.has_nac) { dQGn1 = fma(vec[D + a], twn1[a], dQGn1); piq_n1 = fma(piq, tn1[a], piq_n1); }
                piq_dq = fma(piq, vec[a], piq_dq);
                p0_dQ = fma(cp0[a], vec[D + a], p0_dQ);
                eps = fma(vec[2 * D + a], vec[3 * D + a], eps);
            }
            eps *= -0.5 * ihb * ihb;                                                // (74), b0 = 0
            const cplx yy = form(Y, Y);
            const cplx gamma = c_make(eps - 0.5 * ihb * ihb * yy.x, -0.5 * ihb * ihb * yy.y);   // (84)
            const cplx q_rqq_q = c_sub(c_make(dqCdq, 0), form(UDQ, UDQ));
            const cplx Q_rQQ_Q = c_sub(c_make(dQGdQ, 0), form(WDQ, WDQ));
            const cplx q_rqQ_Q = form(UDQ, WDQ);
            const cplx Pq_dq = c_sub(c_make(piq_dq, 0), form(UDQ, Y));
            const cplx PQ_dQ = c_add(c_make(p0_dQ, 0), form(WDQ, Y));
            cplx ex = gamma;
            ex = c_sub(ex, c_scale(q_rqq_q, 0.5));
            ex = c_sub(ex, c_scale(Q_rQQ_Q, 0.5));
            ex = c_add(ex, q_rqQ_Q);
            ex = c_add(ex, c_mul(c_make(0.0, -ihb), Pq_dq));
            ex = c_add(ex, c_mul(c_make(0.0, ihb), PQ_dQ));
            // branch-tracked square roots
            cplx *prevA = (cplx *)A.wc.detA + tr, *prevM = (cplx *)A.wc.detM + tr;
            const double sA = wm_track_sign(A.track, detA, prevA, A.wc.sgnA + tr);
            const double sM = wm_track_sign(A.track, detM, prevM, A.wc.sgnM + tr);
            const cplx cpre = c_scale(c_sqrt(((const cplx *)A.st.c2)[tr]), A.st.sgn[tr]);
            cplx pre = c_mul(cpre, c_exp(c_make(0.0, A.st.act[tr] * ihb)));
            pre = c_mul(pre, c_scale(c_inv(c_sqrt(detA)), sA));
            pre = c_mul(pre, c_scale(c_inv(c_sqrt(detM)), sM));
            const double w = W.pre / (A.mc_norm * A.probi[tr]);
            const cplx cq = c_scale(c_mul(pre, c_exp(ex)), w);                      // (85) / (n P (2 pi hbar)^D)
            acc[0] += cq.x; acc[1] += cq.y;
            if (A.cq_out) ((cplx *)A.cq_out)[tr] = cq;
            if (W.coef_out) {                   // eqn (75) without its x-dependent part, propagators.py:1408-1432
                cplx v = c_mul(cpre, c_exp(c_make(0.0, A.st.act[tr] * ihb)));
                v = c_mul(v, c_scale(c_inv(c_sqrt(detA)), sA));
                v = c_mul(v, c_exp(c_make(eps - 0.5 * dqCdq, -ihb * piq_dq)));
                ((cplx *)W.coef_out)[tr] = c_scale(v, W.pre_coef / (A.mc_norm * A.probi[tr]));
            }
            if (A.has_nac) {
                const cplx nacqQ = form(UN1, WN1);
                const cplx PQ_n1 = c_add(c_make(tp0n1, 0), form(WN1, Y));
                const cplx Pq_n1 = c_sub(c_make(piq_n1, 0), form(UN1, Y));
                cplx nacQ = c_sub(c_make(dQGn1, 0), form(WDQ, WN1));                // dQ^T RQQ n1
                nacQ = c_sub(nacQ, form(UDQ, WN1));                                 // - dq^T RqQ n1
                nacQ = c_add(nacQ, c_mul(c_make(0.0, -ihb), PQ_n1));
                nacQ.x += tn2Q;
                cplx nacq = c_sub(c_make(dqCn1, 0), form(UDQ, UN1));                // dq^T Rqq n1
                nacq = c_sub(nacq, form(UN1, WDQ));                                 // - n1^T RqQ dQ
                nacq = c_add(nacq, c_mul(c_make(0.0, ihb), Pq_n1));
                nacq.x += tn2q;
                cplx kq = c_mul(c_add(nacqQ, c_mul(nacQ, nacq)), cq);               // (100)
                kq = c_scale(kq, ihb * ihb);
                acc[2] += kq.x; acc[3] += kq.y;
                if (A.kq_out) ((cplx *)A.kq_out)[tr] = kq;
            }
        }
    }
    (void)red;
    if (tid == 0) for (int i = 0; i < 4; ++i) A.partials[(size_t)blockIdx.x * 4 + i] = acc[i];
    // fewer workgroups than partial-sum slots (scratch-limited grid): the first workgroup clears the rest
    if (blockIdx.x == 0)
        for (int i = 4 * (int)gridDim.x + tid; i < 4 * A.npartials; i += nth) A.partials[i] = 0.0;
}

size_t wm_lds_bytes(int D, int dp) {
    const size_t E = 2 * (size_t)dp;
    const size_t doubles = 8 * (size_t)D + 3 * (size_t)D + (D & 1) + wm_region_p(D, dp);
    const size_t cplxs = wm_region_q(D, dp) + (size_t)D * E + (E > (size_t)D ? E : D) + 5 * (size_t)D;
    return doubles * 8 + cplxs * 16 + 32;
}

size_t wm_const_bytes(int D, int dp) { return ((size_t)D * dp + 5 * (size_t)D * D + 5 * (size_t)D) * 8; }

// WM wavefunction on a grid: one workgroup per grid point, threads stride over the trajectories
struct WmGridArgs {
    const double *qp, *coef, *cqq, *dvec;
    int64_t n;
    int D, nx;
    const double *X;
    double *phi;
};

__global__ __launch_bounds__(256) void wm_grid_sum_kernel(WmGridArgs A) {
    extern __shared__ double2 smem2[];
    __shared__ double red[32];
    double *xs = (double *)smem2;
    const int tid = threadIdx.x, D = A.D, k = blockIdx.x;
    for (int a = tid; a < D; a += 256) xs[a] = A.X[(size_t)k * D + a];
    __syncthreads();
    double acc[2] = {0.0, 0.0};
    for (int64_t i = tid; i < A.n; i += 256) {
        const double *Q = A.qp + i * 2 * D;
        const cplx *C = (const cplx *)A.cqq + i * (int64_t)D * D, *d = (const cplx *)A.dvec + i * D;
        cplx ex = c_make(0.0, 0.0);
        for (int a = 0; a < D; ++a) {
            const double dxa = xs[a] - Q[a];
            cplx row = c_make(0.0, 0.0);
            for (int b = 0; b < D; ++b) {
                const double dxb = xs[b] - Q[b];
                row.x = fma(C[a * D + b].x, dxb, row.x); row.y = fma(C[a * D + b].y, dxb, row.y);
            }
            ex.x += dxa * (d[a].x - 0.5 * row.x);
            ex.y += dxa * (d[a].y - 0.5 * row.y);
        }
        const cplx t = c_mul(((const cplx *)A.coef)[i], c_exp(ex));
        acc[0] += t.x; acc[1] += t.y;
    }
    block_sum<2>(acc, red);
    if (tid == 0) { A.phi[2 * (size_t)k] = acc[0]; A.phi[2 * (size_t)k + 1] = acc[1]; }
}

// O(n^2) pair sum behind WaltonManolopoulosPropagator.norm() (reference propagators.py:1484-1575):
//   norm^2 = sum_ij conj(v_i) O_ij v_j,
//   O_ij = det(D'_ij / 2 pi)^(-1/2) exp( -1/2 dQ^T CQQ_j dQ - d_j . dQ + 1/2 b^T D_ij^-1 b ),
//   dQ = Q_j - Q_i,  D_ij = conj(CQQ_i) + CQQ_j,  b = CQQ_j dQ + conj(d_i) + d_j,
// with D_ij inverted in the non-zero subspace: D' = U^T D U (the host passes the projected C'QQ = U^T CQQ U and
// d' = U^T d per trajectory), b^T D^-1 b = b'^T D'^-1 b', b' = U^T b.  One thread per pair (16 x 16 pairs per
// workgroup); the d' x d' system is solved by Gaussian elimination with partial pivoting in per-thread storage.
#define WMN_MAXD 64
#define WMN_MAXDP 16
struct WmPairArgs {
    const double *qp_i, *coef_i, *cqqp_i, *dvecp_i;                       // bras i: [ni] ...
    const double *qp, *coef, *cqq, *dvec, *cqqp, *dvecp, *U;               // kets j: [nj] ...
    int64_t ni, n;
    int D, dp;
    double *partials;
};

__global__ __launch_bounds__(256) void wm_pair_sum_kernel(WmPairArgs A) {
    __shared__ double red[32];
    const int tid = threadIdx.x, D = A.D, dp = A.dp;
    const int64_t tiles = (A.n + 15) / 16;
    const int64_t i = (blockIdx.x / tiles) * 16 + (tid >> 4), j = (blockIdx.x % tiles) * 16 + (tid & 15);
    double acc[2] = {0.0, 0.0};
    if (i < A.ni && j < A.n) {
        const double *Qi = A.qp_i + i * 2 * D, *Qj = A.qp + j * 2 * D;
        const cplx *Cj = (const cplx *)A.cqq + j * (int64_t)D * D, *dj = (const cplx *)A.dvec + j * D;
        const cplx *Cpi = (const cplx *)A.cqqp_i + i * (int64_t)dp * dp, *Cpj = (const cplx *)A.cqqp + j * (int64_t)dp * dp;
        const cplx *dpi = (const cplx *)A.dvecp_i + i * dp, *dpj = (const cplx *)A.dvecp + j * dp;
        cplx w[WMN_MAXD], Dm[WMN_MAXDP * WMN_MAXDP], b[WMN_MAXDP];
        cplx quad = c_make(0, 0), djq = c_make(0, 0);
        for (int a = 0; a < D; ++a) {
            cplx s = c_make(0, 0);
            for (int c = 0; c < D; ++c) {
                const double dq = Qj[c] - Qi[c];
                s.x = fma(Cj[a * D + c].x, dq, s.x); s.y = fma(Cj[a * D + c].y, dq, s.y);
            }
            w[a] = s;
            const double dqa = Qj[a] - Qi[a];
            quad.x = fma(dqa, s.x, quad.x); quad.y = fma(dqa, s.y, quad.y);
            djq.x = fma(dqa, dj[a].x, djq.x); djq.y = fma(dqa, dj[a].y, djq.y);
        }
        for (int k = 0; k < dp; ++k) {
            cplx s = c_make(dpi[k].x + dpj[k].x, -dpi[k].y + dpj[k].y);
            for (int a = 0; a < D; ++a) { const double u = A.U[a * dp + k]; s.x = fma(u, w[a].x, s.x); s.y = fma(u, w[a].y, s.y); }
            b[k] = s;
            for (int l = 0; l < dp; ++l)
                Dm[k * dp + l] = c_make(Cpi[k * dp + l].x + Cpj[k * dp + l].x, -Cpi[k * dp + l].y + Cpj[k * dp + l].y);
        }
        // b^T D'^-1 b and det D' : z = D'^-1 b by elimination on the rows (row operations do not change b^T-form
        // when we keep the ORIGINAL b for the final product)
        cplx z[WMN_MAXDP];
        for (int k = 0; k < dp; ++k) z[k] = b[k];
        cplx det = c_make(1.0, 0.0);
        bool singular = false;
        for (int k = 0; k < dp && !singular; ++k) {
            int piv = k;
            double best = c_abs2(Dm[k * dp + k]);
            for (int r = k + 1; r < dp; ++r) { const double m = c_abs2(Dm[r * dp + k]); if (m > best) { best = m; piv = r; } }
            if (best == 0.0) { singular = true; break; }
            if (piv != k) {
                for (int l = 0; l < dp; ++l) { const cplx t = Dm[k * dp + l]; Dm[k * dp + l] = Dm[piv * dp + l]; Dm[piv * dp + l] = t; }
                const cplx t = z[k]; z[k] = z[piv]; z[piv] = t;
                det = c_make(-det.x, -det.y);
            }
            const cplx p = Dm[k * dp + k], ip = c_inv(p);
            det = c_mul(det, p);
            for (int r = k + 1; r < dp; ++r) {
                const cplx f = c_mul(Dm[r * dp + k], ip);
                for (int l = k + 1; l < dp; ++l) Dm[r * dp + l] = c_fnma(f, Dm[k * dp + l], Dm[r * dp + l]);
                z[r] = c_fnma(f, z[k], z[r]);
            }
        }
        if (!singular) {
            for (int k = dp - 1; k >= 0; --k) {                   // back substitution
                cplx s = z[k];
                for (int l = k + 1; l < dp; ++l) s = c_fnma(Dm[k * dp + l], z[l], s);
                z[k] = c_mul(s, c_inv(Dm[k * dp + k]));
            }
            cplx bib = c_make(0, 0);
            for (int k = 0; k < dp; ++k) bib = c_fma(b[k], z[k], bib);
            double scale = 1.0;
            for (int k = 0; k < dp; ++k) scale *= 1.0 / (2.0 * 3.14159265358979323846);
            const cplx dets = c_scale(det, scale);                 // det(D'/(2 pi))
            const cplx ex = c_make(-0.5 * quad.x - djq.x + 0.5 * bib.x, -0.5 * quad.y - djq.y + 0.5 * bib.y);
            const cplx ol = c_mul(c_inv(c_sqrt(dets)), c_exp(ex));
            const cplx vi = ((const cplx *)A.coef_i)[i], vj = ((const cplx *)A.coef)[j];
            const cplx t = c_mul(c_mul(c_conj(vi), ol), vj);
            acc[0] = t.x; acc[1] = t.y;
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

}  // namespace

extern "C" int64_t sc_wm_pair_sum_tiles(int64_t n) {
    const int64_t t = (n + 15) / 16;
    return t * t;
}
extern "C" int64_t sc_wm_pair_sum_rect_tiles(int64_t ni, int64_t nj) { return ((ni + 15) / 16) * ((nj + 15) / 16); }

extern "C" int sc_wm_pair_sum_rect(const double *qp_i, const double *coef_i, const double *cqqp_i, const double *dvecp_i, int64_t ni,
                                   const double *qp_j, const double *coef_j, const double *cqq_j, const double *dvec_j,
                                   const double *cqqp_j, const double *dvecp_j, int64_t nj, const double *U, int32_t D,
                                   int32_t dprime, double *partials, void *stream) {
    if (!qp_i || !coef_i || !cqqp_i || !dvecp_i || !qp_j || !coef_j || !cqq_j || !dvec_j || !cqqp_j || !dvecp_j || !U || !partials)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_wm_pair_sum: null argument");
    if (D < 1 || D > WMN_MAXD || dprime < 1 || dprime > WMN_MAXDP || dprime > D)
        return sc_fail(SC_ERR_UNSUPPORTED, "sc_wm_pair_sum: D=%d d'=%d outside D <= %d, d' <= %d", D, dprime, WMN_MAXD, WMN_MAXDP);
    if (ni <= 0 || nj <= 0) return SC_OK;
    const int64_t tiles = sc_wm_pair_sum_rect_tiles(ni, nj);
    if (tiles > 0x7fffffff) return sc_fail(SC_ERR_UNSUPPORTED, "sc_wm_pair_sum: %lld x %lld pairs need more than 2^31 tiles", (long long)ni, (long long)nj);
    WmPairArgs a{qp_i, coef_i, cqqp_i, dvecp_i, qp_j, coef_j, cqq_j, dvec_j, cqqp_j, dvecp_j, U, ni, nj, D, dprime, partials};
    hipLaunchKernelGGL(wm_pair_sum_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, a);
    return sc_check_launch("sc_wm_pair_sum");
}

extern "C" int sc_wm_pair_sum(const double *qp, const double *coef, const double *cqq, const double *dvec,
                              const double *cqqp, const double *dvecp, const double *U, int64_t n, int32_t D,
                              int32_t dprime, double *partials, void *stream) {
    return sc_wm_pair_sum_rect(qp, coef, cqqp, dvecp, n, qp, coef, cqq, dvec, cqqp, dvecp, n, U, D, dprime, partials, stream);
}

extern "C" int sc_wm_grid_sum(const double *qp, const double *coef, const double *cqq, const double *dvec, int64_t n,
                              int32_t D, const double *X, int32_t nx, double *phi, void *stream) {
    if (!qp || !coef || !cqq || !dvec || !X || !phi) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_wm_grid_sum: null argument");
    if (D < 1 || D > 512) return sc_fail(SC_ERR_UNSUPPORTED, "sc_wm_grid_sum: D=%d outside 1..512", D);
    if (nx <= 0 || n <= 0) return SC_OK;
    WmGridArgs a{qp, coef, cqq, dvec, n, D, nx, X, phi};
    hipLaunchKernelGGL(wm_grid_sum_kernel, dim3((unsigned)nx), dim3(256), D * sizeof(double), (hipStream_t)stream, a);
    return sc_check_launch("sc_wm_grid_sum");
}

// partial-sum slots: the first wm_main_grid(n) belong to the kernel that processes all trajectories, the last
// WM_RERUN_GRID to the pivoted re-run of the trajectories the register kernel flagged
static const int WM_RERUN_GRID = 256;
static int wm_main_grid(int64_t n) { return (int)(n < 2048 ? (n > 0 ? n : 1) : 2048); }
extern "C" int sc_wm_grid(int64_t n, int32_t dim) {
    (void)dim;
    return wm_main_grid(n) + WM_RERUN_GRID;
}

// workgroups of the global-scratch variant: bounded so that the scratch stays within a few hundred MB
static int wm_scratch_grid(int64_t n) { return (int)(n < 512 ? (n > 0 ? n : 1) : 512); }
static size_t wm_scratch_stride(int D, int dp) { return (wm_lds_bytes(D, dp) + 255) & ~(size_t)255; }
static bool wm_has_small_kernel(int D, int dp) {
    return (D == dp && D >= 1 && D <= 8) || (D == 9 && dp == 3) || (D == 12 && dp == 6);
}

extern "C" int64_t sc_wm_scratch_bytes(int64_t n, int32_t dim, int32_t dprime) {
    if (dim < 1 || dprime < 1 || dprime > dim) return -1;
    // register kernel: the per-trajectory scalars it hands to its tail kernel
    if (wm_has_small_kernel(dim, dprime)) return (int64_t)WM_TAIL_FIELDS * 8 * (n > 0 ? n : 0);
    if (wm_lds_bytes(dim, dprime) <= 160 * 1024) return 0;
    return (int64_t)wm_scratch_stride(dim, dprime) * wm_scratch_grid(n);
}

extern "C" int sc_wm_correlate(const sc_state *st, const sc_wm_consts *wc, const double *zi, const double *probi,
                               double mc_norm, int32_t track, int32_t has_nac, double *cq_out, double *kq_out,
                               double *partials, void *stream) {
    if (!st || !wc || !zi || !probi || !partials) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_wm_correlate: null argument");
    if (wc->dim != st->dim) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_wm_correlate: dimension mismatch");
    if (int rq = sc_require_rowmajor(st, "sc_wm_correlate")) return rq;
    if (wc->dprime < 1 || wc->dprime > st->dim) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_wm_correlate: d' outside 1..D");
    if (st->n <= 0) return SC_OK;
    const int D = st->dim, dp = wc->dprime, grid = wm_main_grid(st->n), slots = sc_wm_grid(st->n, D);
    WmArgs a;
    a.st = *st; a.wc = *wc; a.zi = zi; a.probi = probi; a.mc_norm = mc_norm; a.track = track; a.has_nac = has_nac;
    a.stage_consts = 0; a.cq_out = cq_out; a.kq_out = kq_out; a.partials = partials;
    a.scratch = nullptr; a.scratch_stride = 0; a.npartials = slots; a.only_flagged = nullptr;
    hipStream_t s = (hipStream_t)stream;
    size_t lds = wm_lds_bytes(D, dp);
    // the LDS kernel (every matrix of a trajectory in LDS, full partial pivoting): all trajectories, or only the flagged ones
    auto launch_lds_kernel = [&](int nblocks) {
        a.stage_consts = lds + wm_const_bytes(D, dp) <= 64 * 1024;   // keep >= 2 workgroups per CU
        size_t bytes = lds + (a.stage_consts ? wm_const_bytes(D, dp) : 0);
        if (hipFuncSetAttribute((const void *)wm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
            return sc_check_launch("sc_wm_correlate (LDS attribute)");
        const int threads = D <= 16 ? 64 : 256;
        hipLaunchKernelGGL(wm_kernel<false>, dim3(nblocks), dim3(threads), bytes, s, a);
        return sc_check_launch("sc_wm_correlate");
    };
    // the register kernel eliminates in a fixed pivot order and needs wc->flags to hand weak pivots to the pivoted re-run:
    // without flags (or with per-trajectory coupling vectors) every trajectory takes the fully pivoted general kernel
    if (wm_has_small_kernel(D, dp) && !wc->nac_traj && wc->flags) {
        if (!wc->scratch || wc->scratch_bytes < sc_wm_scratch_bytes(st->n, D, dp))
            return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_wm_correlate: D=%d d'=%d needs a scratch buffer of %lld B (sc_wm_scratch_bytes), "
                           "got %lld", D, dp, (long long)sc_wm_scratch_bytes(st->n, D, dp), (long long)wc->scratch_bytes);
        a.scratch = wc->scratch;
        if (wc->flags && hipMemsetAsync(wc->flags + st->n, 0, sizeof(int32_t), s) != hipSuccess)
            return sc_check_launch("sc_wm_correlate (flag counter)");
        const int rc = sc_wm_launch_small(a, grid, s);
        if (rc < 0) return rc;
        if (rc != 0) {
            // slots of the re-run: written by it (zeros from its idle workgroups), or cleared here when there is none
            a.scratch = nullptr;
            a.partials = partials + 4 * (size_t)grid;
            a.npartials = WM_RERUN_GRID;
            if (!wc->flags) {
                if (hipMemsetAsync(a.partials, 0, 4 * sizeof(double) * WM_RERUN_GRID, s) != hipSuccess)
                    return sc_check_launch("sc_wm_correlate (re-run slots)");
                return SC_OK;
            }
            a.only_flagged = wc->flags;
            return launch_lds_kernel(WM_RERUN_GRID);
        }
    }
    if (lds > 160 * 1024) {
        // the matrices of one trajectory do not fit LDS: run the same kernel on the caller's scratch block
        const int gs = wm_scratch_grid(st->n);
        const size_t stride = wm_scratch_stride(D, dp);
        if (!wc->scratch || wc->scratch_bytes < (int64_t)(stride * gs))
            return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_wm_correlate: D=%d d'=%d needs a scratch buffer of %zu B "
                           "(sc_wm_scratch_bytes), got %lld", D, dp, stride * gs, (long long)wc->scratch_bytes);
        a.scratch = wc->scratch; a.scratch_stride = stride;
        hipLaunchKernelGGL(wm_kernel<true>, dim3(gs), dim3(256), 0, s, a);
        return sc_check_launch("sc_wm_correlate (global scratch)");
    }
    return launch_lds_kernel(grid);
}
