// Fused Herman-Kluk step: RK4 of (q, p, S, monodromy) + HK prefactor + sqrt branch tracking.
//
// One workgroup owns one trajectory at a time (grid-stride over trajectories).  The trajectory's
// four D x D monodromy blocks are contiguous in HBM (engine layout, include/semiclassical_hip.h),
// so every load/store below is a unit-stride sweep.
//
// Reference semantics reproduced (paths relative to the reference repository):
//   RK4 stage formula                       semiclassical/propagators.py:86-119
//   slopes of q, p, Mqq, Mqp, Mpq, Mpp, S   semiclassical/propagators.py:313-383
//   mean of T+V at the k4 stage             semiclassical/propagators.py:380   (quirk Q2)
//   potentials                              semiclassical/potentials.py:63-134, 265-327, 553-593
//   HK prefactor, eqn (29)                  semiclassical/propagators.py:951-1004
//   branch tracking of sqrt(c2)             semiclassical/propagators.py:1006-1052
#include "sc_common.h"
#include "sc_prefactor.h"

namespace {

__device__ __forceinline__ size_t align2(size_t x) { return (x + 1) & ~size_t(1); }

template <bool DENSE>
__global__ __launch_bounds__(256) void hk_step_kernel(StepArgs A) {
    extern __shared__ double2 smem2[];
    double *smem = (double *)smem2;
    const int D = A.st.dim, DD = D * D, tid = threadIdx.x, nth = blockDim.x;
    const int dp = A.hk.dprime;
    const bool diag = A.hk.diag != 0;
    // mode & 0xff: 0 = step + prefactor, 1 = prefactor only + tracker initialisation
    // mode & 0x200: fix-up pass -- only trajectories flagged by the fast path, prefactor + tracking only
    const bool fixup = (A.mode & 0x200) != 0;
    if (fixup && A.st.flags[A.st.n] == 0) return;          // flags[n] counts the trajectories flagged in this step
    const bool do_step = (A.mode & 0xff) == 0 && !fixup;
    const bool init_track = (A.mode & 0xff) == 1;
    const double dt = A.dt, hh = 0.5 * dt, h6 = dt / 6.0;

    // ---- LDS carve-up (all offsets even => 16-byte aligned) ----
    size_t off = 0;
    double *red = smem + off;  off += 32;
    int *ipiv = (int *)(smem + off); off += 2;
    double *hst = smem + off;  off += align2(4 * D);     // stage Hessian diagonals (separable)
    double *vec = smem + off;  off += align2(4 * D);     // st, 1/st, si, 1/si (diag prefactor)
    cplx *mat = (cplx *)(smem + off); off += 2 * (size_t)dp * dp;
    cplx *X = (cplx *)(smem + off);
    if (!diag) off += 2 * (size_t)D * dp;
    // dense potential: Mq, Mp, Aq, Ap, Qa, Qb, Ps (each D x 2D), hess (D x D), dr (D)
    double *Mq = smem + off, *Mp = Mq + 2 * DD, *Aq = Mp + 2 * DD, *Ap = Aq + 2 * DD;
    double *Qa = Ap + 2 * DD, *Qb = Qa + 2 * DD, *Ps = Qb + 2 * DD;
    double *Hm = Ps + 2 * DD, *drv = Hm + DD;

    if (diag) {
        for (int a = tid; a < D; a += nth) {
            vec[a] = A.hk.st[a]; vec[D + a] = 1.0 / A.hk.st[a];
            vec[2 * D + a] = A.hk.si[a]; vec[3 * D + a] = 1.0 / A.hk.si[a];
        }
    }
    if (DENSE) {
        for (int e = tid; e < DD; e += nth) Hm[e] = A.pot.par2[e];
    }
    __syncthreads();

    double esum = 0.0;
    for (int64_t tr = blockIdx.x; tr < A.st.n; tr += gridDim.x) {
        if (fixup && A.st.flags[tr] == 0) continue;          // uniform over the workgroup
        double *qp = A.st.qp + tr * 2 * D;
        double *M = A.st.mono + tr * 4 * (int64_t)DD;
        double im = 1.0;

        if (do_step) {
            // ---------------- phase A: classical trajectory (q, p, S) ----------------
            double red5[5] = {0, 0, 0, 0, 0};
            double qn = 0, pn = 0;
            const bool own = tid < D;
            double q = 0, p = 0;
            if (own) { q = qp[tid]; p = qp[D + tid]; im = A.pot.inv_mass[tid]; }
            if (!DENSE) {
                if (own) {
                    const double c0 = A.pot.par0[tid], c1 = A.pot.par1 ? A.pot.par1[tid] : 0.0;
                    double v, g, h;
                    sep_eval(A.pot.kind, c0, c1, q, v, g, h);
                    const double kq1 = p * im, kp1 = -g; hst[tid] = h;
                    red5[0] = 0.5 * p * p * im - v;
                    const double q2 = q + hh * kq1, p2 = p + hh * kp1;
                    sep_eval(A.pot.kind, c0, c1, q2, v, g, h);
                    const double kq2 = p2 * im, kp2 = -g; hst[D + tid] = h;
                    red5[1] = 0.5 * p2 * p2 * im - v;
                    const double q3 = q + hh * kq2, p3 = p + hh * kp2;
                    sep_eval(A.pot.kind, c0, c1, q3, v, g, h);
                    const double kq3 = p3 * im, kp3 = -g; hst[2 * D + tid] = h;
                    red5[2] = 0.5 * p3 * p3 * im - v;
                    const double q4 = q + dt * kq3, p4 = p + dt * kp3;
                    sep_eval(A.pot.kind, c0, c1, q4, v, g, h);
                    const double kq4 = p4 * im, kp4 = -g; hst[3 * D + tid] = h;
                    red5[3] = 0.5 * p4 * p4 * im - v;
                    red5[4] = 0.5 * p4 * p4 * im + v;
                    qn = q + h6 * (kq1 + 2.0 * kq2 + 2.0 * kq3 + kq4);
                    pn = p + h6 * (kp1 + 2.0 * kp2 + 2.0 * kp3 + kp4);
                }
            } else {
                // V = E0 + g.dr + 1/2 dr.H.dr - origin ; grad = g + H.dr      potentials.py:583-590
                double qs = q, ps = p, kqs = 0, kps = 0;
                const double g0 = own ? A.pot.par1[tid] : 0.0, x0 = own ? A.pot.par0[tid] : 0.0;
                for (int s = 0; s < 4; ++s) {
                    if (s > 0) { const double c = (s == 3) ? dt : hh; qs = q + c * kqs; ps = p + c * kps; }
                    __syncthreads();
                    if (own) drv[tid] = qs - x0;
                    __syncthreads();
                    double kq = 0, kp = 0;
                    if (own) {
                        double hd = 0.0;
                        for (int b = 0; b < D; ++b) hd = fma(Hm[tid * D + b], drv[b], hd);
                        const double dr = qs - x0;
                        const double v = dr * g0 + 0.5 * dr * hd;   // + scalar0 added after the reduction
                        kq = ps * im; kp = -(g0 + hd);
                        const double t = 0.5 * ps * ps * im;
                        red5[s] = t - v;
                        if (s == 3) red5[4] = t + v;
                        const double w = (s == 0 || s == 3) ? 1.0 : 2.0;
                        qn += w * kq; pn += w * kp;
                    }
                    kqs = kq; kps = kp;
                }
                if (own) { qn = q + h6 * qn; pn = p + h6 * pn; }
            }
            block_sum<5>(red5, red);
            if (DENSE) {
                for (int s = 0; s < 4; ++s) red5[s] -= A.pot.scalar0;
                red5[4] += A.pot.scalar0;
            }
            if (own) { qp[tid] = qn; qp[D + tid] = pn; }
            if (tid == 0) {
                A.st.act[tr] += h6 * (red5[0] + 2.0 * red5[1] + 2.0 * red5[2] + red5[3]);
                esum += red5[4];
            }
            __syncthreads();   // hst visible
        }

        // ---------------- phase B: monodromy blocks ----------------
        if (!DENSE) {
            const int lay = A.st.mono_layout;       // tiled only in the fix-up pass behind the fast path
            for (int e = tid; e < DD; e += nth) {
                const int a = e / D, b = e - a * D;
                double mqq, mqp, mpq, mpp;
                if (lay == SC_MONO_ROWMAJOR) { mqq = M[e]; mqp = M[DD + e]; mpq = M[2 * DD + e]; mpp = M[3 * DD + e]; }
                else {
                    mqq = M[sc_mono_offset(lay, D, 0, a, b)]; mqp = M[sc_mono_offset(lay, D, 1, a, b)];
                    mpq = M[sc_mono_offset(lay, D, 2, a, b)]; mpp = M[sc_mono_offset(lay, D, 3, a, b)];
                }
                if (do_step) {
                    const double ima = A.pot.inv_mass[a];
                    const double h1 = hst[a], h2 = hst[D + a], h3 = hst[2 * D + a], h4 = hst[3 * D + a];
                    rk4_pair(mqq, mpq, ima, h1, h2, h3, h4, dt);
                    rk4_pair(mqp, mpp, ima, h1, h2, h3, h4, dt);
                    M[e] = mqq; M[DD + e] = mqp; M[2 * DD + e] = mpq; M[3 * DD + e] = mpp;
                }
                if (diag) {
                    const double sta = vec[a], ista = vec[D + a], sib = vec[2 * D + b], isib = vec[3 * D + b];
                    mat[e] = c_make(0.5 * (sta * isib * mqq + ista * sib * mpp),
                                    0.5 * (-SC_HBAR * sta * sib * mqp + (1.0 / SC_HBAR) * ista * isib * mpq));
                }
            }
            __syncthreads();
            if (!diag) general_prefactor_matrix(A.hk, M, M + DD, M + 2 * DD, M + 3 * DD, D, X, mat);
        } else {
            const int W = 2 * D;
            for (int e = tid; e < DD; e += nth) {
                const int a = e / D, b = e - a * D;
                Mq[a * W + b] = M[e]; Mq[a * W + D + b] = M[DD + e];
                Mp[a * W + b] = M[2 * DD + e]; Mp[a * W + D + b] = M[3 * DD + e];
            }
            __syncthreads();
            if (do_step) {
                const double *Qc = Mq;
                double *Qn = Qa;
                for (int s = 0; s < 4; ++s) {
                    const double w = (s == 0 || s == 3) ? 1.0 : 2.0;
                    const double c = (s == 2) ? dt : hh;
                    for (int e = tid; e < 2 * DD; e += nth) {
                        const int a = e / W, col = e - a * W;
                        const double pc = (s == 0) ? Mp[e] : Ps[e];
                        const double kq = pc * A.pot.inv_mass[a];
                        double hq = 0.0;
                        for (int g = 0; g < D; ++g) hq = fma(Hm[a * D + g], Qc[g * W + col], hq);
                        const double kp = -hq;
                        if (s == 0) { Aq[e] = kq; Ap[e] = kp; } else { Aq[e] += w * kq; Ap[e] += w * kp; }
                        if (s < 3) { Qn[e] = Mq[e] + c * kq; Ps[e] = Mp[e] + c * kp; }
                    }
                    __syncthreads();
                    Qc = Qn;
                    Qn = (Qn == Qa) ? Qb : Qa;
                }
                for (int e = tid; e < 2 * DD; e += nth) {
                    Mq[e] = Mq[e] + h6 * Aq[e];
                    Mp[e] = Mp[e] + h6 * Ap[e];
                }
                __syncthreads();
                for (int e = tid; e < DD; e += nth) {
                    const int a = e / D, b = e - a * D;
                    M[e] = Mq[a * W + b]; M[DD + e] = Mq[a * W + D + b];
                    M[2 * DD + e] = Mp[a * W + b]; M[3 * DD + e] = Mp[a * W + D + b];
                }
            }
            if (diag) {
                for (int e = tid; e < DD; e += nth) {
                    const int a = e / D, b = e - a * D;
                    const double sta = vec[a], ista = vec[D + a], sib = vec[2 * D + b], isib = vec[3 * D + b];
                    mat[e] = c_make(0.5 * (sta * isib * Mq[a * W + b] + ista * sib * Mp[a * W + D + b]),
                                    0.5 * (-SC_HBAR * sta * sib * Mq[a * W + D + b]
                                           + (1.0 / SC_HBAR) * ista * isib * Mp[a * W + b]));
                }
                __syncthreads();
            } else {
                general_prefactor_matrix(A.hk, Mq, Mq + D, Mp, Mp + D, W, X, mat);
            }
        }

        // ---------------- phase C: c2 = det(mat), branch tracking ----------------
        const cplx det = lds_lu_det(mat, dp, ipiv);
        if (tid == 0) {
            cplx *c2 = (cplx *)A.st.c2;
            if (!init_track) {
                const cplx prev = c2[tr];
                if (prev.x < 0.0 && det.x < 0.0 && prev.y * det.y < 0.0) A.st.sgn[tr] = -A.st.sgn[tr];
            } else {
                A.st.sgn[tr] = 1.0;
            }
            c2[tr] = det;
            if (fixup) A.st.flags[tr] = 0;
        }
        __syncthreads();
    }
    if (tid == 0 && A.epart && !fixup) A.epart[blockIdx.x] = esum;
}

}  // namespace

int sc_launch_step_sd(const StepArgs &a, hipStream_t s);   // sc_hk_step_sd.hip
#ifdef SC_TUNING
int sc_launch_step_rw(const StepArgs &a, hipStream_t s);   // tools/variants/sc_hk_step_rw.hip
#endif

static int step_threads(int D) { return D * D <= 256 ? 64 : 256; }

extern "C" int sc_step_grid(int64_t n, int32_t dim) {
    int64_t cap = (step_threads(dim) == 64) ? 256 * 16 : 256 * 4;
    return (int)(n < cap ? (n > 0 ? n : 1) : cap);
}

extern "C" int sc_hk_step(const sc_potential *pot, const sc_state *st, const sc_hk_consts *hk, double dt,
                          int32_t mode, double *energy_partials, void *stream) {
    if (!pot || !st || !hk) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_step: null argument");
    const int D = st->dim;
    if (pot->dim != D || hk->dim != D) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_step: dimension mismatch");
    if (D < 1 || D > 64 || hk->dprime < 1 || hk->dprime > D)
        return sc_fail(SC_ERR_UNSUPPORTED, "sc_hk_step: D=%d d'=%d outside 1..64", D, hk->dprime);
    if (hk->diag && hk->dprime != D) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_step: diag prefactor needs d' == D");
    if (st->n <= 0) return SC_OK;
    const bool dense = pot->kind == SC_POT_HARMONIC_DENSE;
    if (!dense && pot->kind != SC_POT_MORSE && pot->kind != SC_POT_HARMONIC_SEP && pot->kind != SC_POT_EPS_MORSE)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_step: unknown potential kind %d", pot->kind);
    bool fast = !dense && hk->diag && st->work;
    // The register kernels for D <= 12 and 16 < D <= 64 eliminate in a fixed (block-)pivot order and hand weak pivots to
    // the fully pivoted fix-up launch THROUGH st->flags.  Without the flag array there is no fix-up, and a zero leading
    // pivot would end as inf / NaN in c2: such callers get the fully pivoted LDS kernel for every trajectory instead.
    // (13 <= D <= 16: hk_step_w16_kernel pivots over the whole row by itself.)
    if (!st->flags && !(D > SC_SEP16_MAX_D && D <= 16)) fast = false;
    if (st->mono_layout != SC_MONO_ROWMAJOR && st->mono_layout != SC_MONO_TILED16)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_step: unknown mono_layout %d", st->mono_layout);
    if (st->mono_layout == SC_MONO_TILED16 && D > 16 && !fast)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_step: the tiled monodromy layout is only taken on the separable / "
                       "diagonal-width fast path (sc_mono_convert the state first)");
    int dbg = 0;
#ifdef SC_TUNING   // experiment knobs exist only in the tuning build (tools/mkvar.sh); the product library reads no environment
    if (getenv("SC_FORCE_GENERAL_STEP")) fast = false;
    dbg = getenv("SC_DEBUG_SKIP_LU") ? 0x100 : 0;      // the fix-up launch is skipped (the ablated kernel is a variant library, SC_SD_ABLATE_LU)
#endif
    if (fast) {
        StepArgs a{*pot, *st, *hk, dt, mode | dbg, energy_partials, sc_step_grid(st->n, st->dim)};
#ifdef SC_TUNING
        // SC_FAST_KERNEL=rw: the rejected row-wave layout of tools/variants/sc_hk_step_rw.hip
        const char *which = getenv("SC_FAST_KERNEL");
        if (which && which[0] == 'r') return sc_launch_step_rw(a, (hipStream_t)stream);
#endif
        // 13 <= D <= 16: hk_step_w16_kernel pivots over the whole row (no flags, no cursor, no fix-up launch); D <= 12:
        // hk_step_sep16_kernel eliminates in a fixed order and flags weak pivots like the 256-thread kernel
        bool small = D > SC_SEP16_MAX_D && D <= 16;
#ifdef SC_TUNING
        if (getenv("SC_NO_WAVE_KERNEL")) small = false;      // the 256-thread kernel is forced: it needs flags and cursor
#endif
        if (!(mode & 0x400)) {      // 0x400 (internal, sc_hk_step_multi): the fast kernels have run, only the fix-up is wanted
            // flags[n]: trajectories flagged in this step, flags[n + 1]: trajectory cursor of the fast kernel
            if (!small && st->flags && hipMemsetAsync(st->flags + st->n, 0, 2 * sizeof(int32_t), (hipStream_t)stream) != hipSuccess)
                return sc_check_launch("sc_hk_step (flag counter)");
            const int rc = sc_launch_step_sd(a, (hipStream_t)stream);
            if (rc != SC_OK || !st->flags || small || (dbg & 0x100)) return rc;
        }
        mode = (mode & ~0x400) | 0x200;      // fully pivoted fix-up of the trajectories the fast path flagged (normally none)
    }
    const size_t DD = (size_t)D * D, dp = hk->dprime;
    size_t doubles = 32 + 2 + 2 * ((4 * D + 1) & ~1) + 2 * dp * dp + (hk->diag ? 0 : 2 * D * dp);
    if (dense) doubles += 7 * 2 * DD + DD + D;
    const size_t lds = doubles * sizeof(double) + 16;
    if (lds > 160 * 1024) return sc_fail(SC_ERR_UNSUPPORTED, "sc_hk_step: needs %zu B of LDS (D=%d)", lds, D);
    StepArgs a{*pot, *st, *hk, dt, mode, energy_partials, sc_step_grid(st->n, st->dim)};
    const int threads = step_threads(D), grid = sc_step_grid(st->n, D);
    hipStream_t s = (hipStream_t)stream;
    // constant Hessian, D <= 16 and the step matrix Phi(dt) at hand: the register kernel of sc_hk_step_lin.hip
    if (dense && D <= 16 && st->flags && (mode == 1 || (mode == 0 && pot->lin_prop && pot->lin_dt == dt))) {
        if (st->flags && hipMemsetAsync(st->flags + st->n, 0, sizeof(int32_t), s) != hipSuccess)
            return sc_check_launch("sc_hk_step (flag counter)");
        const int rc = sc_launch_step_lin(a, grid, s);
        if (rc < 0) return rc;
        if (rc != 0) {
            if (!st->flags) return SC_OK;
            // determinants the fixed pivot order was too weak for (normally none): fully pivoted fix-up, same stream
            a.mode = mode | 0x200;
            a.epart = nullptr;
            if (hipFuncSetAttribute((const void *)hk_step_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return sc_check_launch("sc_hk_step (LDS attribute)");
            hipLaunchKernelGGL(hk_step_kernel<true>, dim3(grid), dim3(threads), lds, s, a);
            return sc_check_launch("sc_hk_step (fix-up)");
        }
    }
    if (dense) {
        if (hipFuncSetAttribute((const void *)hk_step_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return sc_check_launch("sc_hk_step (LDS attribute)");
        hipLaunchKernelGGL(hk_step_kernel<true>, dim3(grid), dim3(threads), lds, s, a);
    } else {
        if (hipFuncSetAttribute((const void *)hk_step_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return sc_check_launch("sc_hk_step (LDS attribute)");
        hipLaunchKernelGGL(hk_step_kernel<false>, dim3(grid), dim3(threads), lds, s, a);
    }
    return sc_check_launch("sc_hk_step");
}

// ---- two time steps per visit (include/semiclassical_hip.h: sc_hk_step_multi) ----
extern "C" int sc_hk_step_multi_supported(const sc_potential *pot, const sc_state *st, const sc_hk_consts *hk) {
    if (!pot || !st || !hk) return 0;
    const bool sep = pot->kind == SC_POT_MORSE || pot->kind == SC_POT_HARMONIC_SEP || pot->kind == SC_POT_EPS_MORSE;
    return sep && hk->diag && hk->dprime == st->dim && pot->dim == st->dim && hk->dim == st->dim && st->dim > 16 && st->dim <= 64 &&
           st->mono_layout == SC_MONO_TILED16 && st->work && st->flags ? 1 : 0;
}

extern "C" int sc_hk_step_multi(const sc_potential *pot, const sc_state *st, const sc_hk_consts *hk, const sc_multi_scratch *ms,
                                double dt, double *energy_partials, void *stream) {
    if (!pot || !st || !hk || !ms) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_step_multi: null argument");
    if (!ms->work || !ms->qp_mid || !ms->act_mid || !ms->c2_mid || !ms->sgn_mid || !ms->unrepaired)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_step_multi: null scratch field");
    if (!sc_hk_step_multi_supported(pot, st, hk))
        return sc_fail(SC_ERR_UNSUPPORTED, "sc_hk_step_multi: needs a separable potential, diagonal width matrices, 16 < D <= 64, the tiled "
                       "storage order (sc_mono_convert) and sc_state.work / flags (use sc_hk_step)");
    if (st->n <= 0) return SC_OK;
    hipStream_t s = (hipStream_t)stream;
    // flags[n]: trajectories flagged in the LAST sub-step, flags[n + 1]: trajectory cursor of the block kernel
    if (hipMemsetAsync(st->flags + st->n, 0, 2 * sizeof(int32_t), s) != hipSuccess) return sc_check_launch("sc_hk_step_multi (flag counter)");
    const int grid = sc_step_grid(st->n, st->dim);
    StepArgs a{*pot, *st, *hk, dt, 0, energy_partials, 2 * grid};
    int rc = sc_launch_step_sd_multi(a, *ms, s);
    if (rc != SC_OK) return rc;
    // weak in-block pivots of the LAST sub-step (normally none): fully pivoted fix-up from the final blocks, as in sc_hk_step
    return sc_hk_step(pot, st, hk, dt, 0x400, nullptr, stream);
}
