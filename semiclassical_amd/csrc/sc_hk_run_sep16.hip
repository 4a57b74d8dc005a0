// The caller loop of the reference (cli.py:401-436: autocorrelation, ic_correlation, step -- nt times) as ONE launch, for
// SEPARABLE potentials with diagonal width matrices and D <= 12 (BASELINE.json configs[0]: the 5-mode anharmonic AS model).
//
// Step by step the loop is six launches per time step (correlate, reduce, step, guard, ...): at n = 1000 the kernels need
// 12 us of a 33 us step, and replaying the sequence from a HIP graph is no faster than issuing it.  Here a trajectory is
// loaded ONCE into the registers of its 16-lane row (the layout of hk_step_sep16_kernel: lane a holds mode a and row a of
// the four monodromy blocks), runs all nsteps steps there and is written back once; between steps nothing touches HBM
// but the five partial sums of the step.  Trajectories are independent, so there is no grid-wide synchronisation: every
// wavefront adds its four trajectories' terms of step k into its own slot partials[k][slot] (one writer per slot: the
// sums are deterministic), and hk_run_reduce_kernel adds the slots of a step in a fixed order afterwards.
//
// Per step and trajectory, in this order (= the order of the caller loop):
//   terms of C_auto and k_ic from the CURRENT state       hk_correlate_kernel's arithmetic, sums over the modes by the same
//                                                         rotate-and-add tree (propagators.py:784-843, 845-911)
//   RK4 of (q_a, p_a), action, <T+V> at the k4 stage,     hk_step_sep16_kernel's arithmetic (propagators.py:86-119, 313-383)
//   row propagators P_a applied to the monodromy rows
//   prefactor row, determinant, branch tracker            fixed pivot order; a weak pivot (never for the diagonal blocks a
//                                                         separable potential produces from M(0) = 1) repeats the
//                                                         elimination with the pivot searched among the lanes
//                                                         (propagators.py:969-1052)
#include "sc_hk_run.h"
#include "sc_row16.h"

namespace {

// value of `v` in lane 0 of each 16-lane row, summed over the four rows of the wavefront (result in every lane)
__device__ __forceinline__ double sum_row_heads(double v, bool head) { return wave_sum(head ? v : 0.0); }

#ifndef SC_RUN_OCC
#define SC_RUN_OCC 2
#endif
template <int DP, int KIND>
__global__ __launch_bounds__(256, SC_RUN_OCC) void hk_run_sep16_kernel(RunArgs R) {
    const StepArgs &A = R.step;
    const int D = A.st.dim, DD = D * D, tid = threadIdx.x, lane = tid & 63, r = tid & 15, grp = tid >> 4, wave = tid >> 6;
    const int rowbase = tid & 48;
    const double dt = A.dt, hh = 0.5 * dt, h6 = dt / 6.0;
    const bool mine = r < D;
    const double sta = mine ? A.hk.st[r] : 1.0, ista = 1.0 / sta;
    const double im = mine ? A.pot.inv_mass[r] : 0.0, c0 = mine ? A.pot.par0[r] : 0.0;
    const double c1 = (mine && A.pot.par1) ? A.pot.par1[r] : 0.0;
    const double ocA = mine ? R.oc.A[r] : 0.0, ocB = mine ? R.oc.B[r] : 0.0, ocC = mine ? R.oc.C[r] : 0.0;
    const double qk = mine ? R.oc.qk[r] : 0.0, pk = mine ? R.oc.pk[r] : 0.0;
    const bool nac = R.has_nac != 0;
    const double nq0 = (mine && nac) ? R.nc.q0[r] : 0.0, np0 = (mine && nac) ? R.nc.p0[r] : 0.0;
    const double nrn = (mine && nac) ? R.nc.rn[r] : 0.0, ngn = (mine && nac) ? R.nc.gn[r] : 0.0;
    kptr ksi = (kptr)A.hk.si;
    const int64_t n = A.st.n, stride = (int64_t)gridDim.x * 16;
    const int slot = blockIdx.x * 4 + wave;
    double one = 1.0;
    asm volatile("" : "+v"(one));

    for (int64_t t0 = (int64_t)blockIdx.x * 16; t0 < n; t0 += stride) {
        const bool active = t0 + grp < n, head = active && r == 0;
        const int64_t tr = active ? t0 + grp : n - 1;          // idle rows shadow the last trajectory and contribute nothing
        double *M = A.st.mono + tr * 4 * (int64_t)DD, *qp = A.st.qp + tr * 2 * D;
        // ---- the trajectory: rows of the four blocks, (q_a, p_a), S, c2, sign, and its time-independent factors
        double mqq[DP], mqp[DP], mpq[DP], mpp[DP];
#pragma unroll
        for (int b = 0; b < DP; ++b) {
            const bool ok = mine && b < D;
            mqq[b] = ok ? M[r * D + b] : 0.0; mqp[b] = ok ? M[DD + r * D + b] : 0.0;
            mpq[b] = ok ? M[2 * DD + r * D + b] : 0.0; mpp[b] = ok ? M[3 * DD + r * D + b] : 0.0;
        }
        double q = mine ? qp[r] : 0.0, p = mine ? qp[D + r] : 0.0;
        double S = A.st.act[tr], sgn = A.st.sgn[tr];
        cplx c2 = ((const cplx *)A.st.c2)[tr];
        const cplx vi = ((const cplx *)R.vi)[tr];
        const cplx nacq = nac ? ((const cplx *)R.nacq)[tr] : c_make(0.0, 0.0);
        const double wgt = 1.0 / (R.mc_norm * R.probi[tr]);

        for (int k = 0; k < R.nsteps; ++k) {
            // ---- terms of the correlation functions from the current state (hk_correlate_kernel, diagonal widths) ----
            double v5[5];
            {
                const double dq = qk - q, dpp = pk - p;
                const double sA = row_sum(dq * ocA * dq), sB = row_sum(dpp * ocB * dpp), sP = row_sum(pk * dq), sC = row_sum(dq * ocC * dpp);
                double sR = 0.0, sG = 0.0;
                if (nac) { sR = row_sum((nq0 - q) * nrn); sG = row_sum((p - np0) * ngn); }
                const cplx ex = c_make(-0.5 * sA - 0.5 / (SC_HBAR * SC_HBAR) * sB, (-sP + sC) / SC_HBAR);
                const cplx vt = c_scale(c_exp(ex), R.oc.fac);
                const cplx c = c_scale(c_sqrt(c2), sgn);
                const cplx ph = c_exp(c_make(0.0, S / SC_HBAR));
                cplx cq = c_mul(c_mul(c_conj(vt), vi), c_mul(c, ph));
                cq = c_scale(cq, wgt);
                v5[0] = cq.x; v5[1] = cq.y; v5[2] = 0.0; v5[3] = 0.0;
                if (nac) {
                    const cplx nacQ = c_make(R.nc.n2 + sR, -(R.nc.p0n1 + sG) / SC_HBAR);
                    cplx kq = c_mul(c_mul(nacQ, nacq), cq);
                    kq = c_scale(kq, 1.0 / (SC_HBAR * SC_HBAR));
                    v5[2] = kq.x; v5[3] = kq.y;
                }
            }
            // ---- RK4 of (q_a, p_a) with the reference's stage formula, action, <T+V> and the row propagator P_a ----
            double red5[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, p11 = 1.0, p12 = 0.0, p21 = 0.0, p22 = 1.0;
            if (mine) {
                double v, g, h1, h2, h3, h4;
                sep_eval(KIND, c0, c1, q, v, g, h1);
                const double kq1 = p * im, kp1 = -g;
                red5[0] = 0.5 * p * p * im - v;
                const double q2 = q + hh * kq1, p2 = p + hh * kp1;
                sep_eval(KIND, c0, c1, q2, v, g, h2);
                const double kq2 = p2 * im, kp2 = -g;
                red5[1] = 0.5 * p2 * p2 * im - v;
                const double q3 = q + hh * kq2, p3 = p + hh * kp2;
                sep_eval(KIND, c0, c1, q3, v, g, h3);
                const double kq3 = p3 * im, kp3 = -g;
                red5[2] = 0.5 * p3 * p3 * im - v;
                const double q4 = q + dt * kq3, p4 = p + dt * kp3;
                sep_eval(KIND, c0, c1, q4, v, g, h4);
                const double kq4 = p4 * im, kp4 = -g;
                red5[3] = 0.5 * p4 * p4 * im - v;
                red5[4] = 0.5 * p4 * p4 * im + v;
                q = q + h6 * (kq1 + 2.0 * kq2 + 2.0 * kq3 + kq4);
                p = p + h6 * (kp1 + 2.0 * kp2 + 2.0 * kp3 + kp4);
                rk4_pair(p11, p21, im, h1, h2, h3, h4, dt);       // (u, v) = (1, 0) -> first column of P_a
                rk4_pair(p12, p22, im, h1, h2, h3, h4, dt);       // (0, 1) -> second column
            }
            {
                double s5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
                dpp_guard(red5);
                sfor<0, DP>([&](auto kc) {            // sums over the lanes k < DP (lanes beyond D hold zeros): as hk_step_sep16_kernel
#pragma unroll
                    for (int i = 0; i < 5; ++i) fmac_bc<decltype(kc)::value>(s5[i], red5[i], one);
                });
                S += h6 * (s5[0] + 2.0 * s5[1] + 2.0 * s5[2] + s5[3]);
                v5[4] = s5[4];
            }
            // ---- (Mqq, Mpq)' = P_a (Mqq, Mpq), (Mqp, Mpp)' = P_a (Mqp, Mpp); prefactor row a (rows and columns beyond D: identity)
            cplx mat[DP], keep[DP];
#pragma unroll
            for (int b = 0; b < DP; ++b) {
                const double nqq = fma(p12, mpq[b], p11 * mqq[b]), npq = fma(p22, mpq[b], p21 * mqq[b]);
                const double nqp = fma(p12, mpp[b], p11 * mqp[b]), npp = fma(p22, mpp[b], p21 * mqp[b]);
                mqq[b] = nqq; mpq[b] = npq; mqp[b] = nqp; mpp[b] = npp;
                const double sib = b < D ? ksi[b] : 1.0, isib = 1.0 / sib;
                mat[b] = (mine && b < D) ? c_make(0.5 * (sta * isib * nqq + ista * sib * npp),
                                                  0.5 * (-SC_HBAR * sta * sib * nqp + (1.0 / SC_HBAR) * ista * isib * npq))
                                         : c_make(r == b ? 1.0 : 0.0, 0.0);
                keep[b] = mat[b];
            }
            int weak = 0;
            cplx det = det_rows_fixed_order<DP>(mat, r, weak);
            if (__builtin_amdgcn_readfirstlane(wave_max_i32(weak)) != 0) {
                // some trajectory of this wavefront met a weak pivot: its determinant by elimination with the pivot searched among
                // the lanes (ds_bpermute), as round 2 did for every trajectory
                int myk, src;
                cplx det2, dummy[1] = {c_make(0.0, 0.0)};
                gauss_jordan_rows<DP, 1>(keep, dummy, r >= DP, r, rowbase, myk, src, det2);
                if (weak) det = det2;
            }
            if (c2.x < 0.0 && det.x < 0.0 && c2.y * det.y < 0.0) sgn = -sgn;       // branch tracker (propagators.py:1045-1047)
            c2 = det;
            // ---- this wavefront's share of step k: one writer per slot ----
#pragma unroll
            for (int i = 0; i < 5; ++i) v5[i] = sum_row_heads(v5[i], head);
            if (lane == 0) {
                double *pp = R.partials + ((size_t)k * R.slots + slot) * 5;
#pragma unroll
                for (int i = 0; i < 5; ++i) __builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double *)(pp + i), v5[i]);
            }
        }
        // ---- the trajectory goes back
        if (active) {
            if (mine) {
#pragma unroll
                for (int b = 0; b < DP; ++b) {
                    if (b < D) {
                        M[r * D + b] = mqq[b]; M[DD + r * D + b] = mqp[b];
                        M[2 * DD + r * D + b] = mpq[b]; M[3 * DD + r * D + b] = mpp[b];
                    }
                }
                qp[r] = q; qp[D + r] = p;
            }
            if (r == 0) {
                A.st.act[tr] = S; A.st.sgn[tr] = sgn;
                ((cplx *)A.st.c2)[tr] = c2;
            }
        }
    }
}

// slots of one step added in a fixed order -> out[k][0..3], mean <T+V> of the step -> out[k][4]
__global__ __launch_bounds__(256) void hk_run_reduce_kernel(const double *partials, int slots, double n_traj, double *out) {
    __shared__ double red[32];
    const int k = blockIdx.x;
    double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < slots; i += 256) {
        const double *pp = partials + ((size_t)k * slots + i) * 5;
#pragma unroll
        for (int j = 0; j < 5; ++j) v[j] += pp[j];
    }
    block_sum<5>(v, red);
    if (threadIdx.x < 5) out[(size_t)k * 5 + threadIdx.x] = threadIdx.x < 4 ? v[threadIdx.x] : v[4] / n_traj;
}

// the energy guard over the steps of a fused run, in order (propagators.py:385-398; sc_energy_guard step by step)
__global__ void hk_run_guard_kernel(const double *out, int nsteps, double *elog) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double prev = elog[1], worst = elog[2], count = elog[3], last = elog[0];
    for (int k = 0; k < nsteps; ++k) {
        const double mean = out[(size_t)k * 5 + 4];
        last = prev;
        if (count >= 1.0) { const double change = fabs(mean - prev); if (change > worst) worst = change; }
        prev = mean; count += 1.0;
    }
    elog[0] = last; elog[1] = prev; elog[2] = worst; elog[3] = count;
}

}  // namespace

extern "C" int sc_hk_run_slots(int64_t n, int32_t dim) {
    (void)dim;
    const int64_t groups = (n + 15) / 16;
    return 4 * (int)(groups < 1024 ? (groups > 0 ? groups : 1) : 1024);
}

// constant dense Hessian with its step matrix, D <= 16 at the shapes sc_hk_run_lin.hip instantiates (real L, R)
static bool run_lin_shape(const sc_potential *pot, const sc_hk_consts *hk, bool modal = false) {
    if (pot->kind != SC_POT_HARMONIC_DENSE || (!modal && !pot->lin_prop) || pot->dim > 16 || hk->dim != pot->dim) return false;
    static const double some = 0.0;
    RunArgs probe{};
    probe.step.st.dim = pot->dim;
    probe.step.hk = *hk;
    probe.mode_prop = modal ? &some : nullptr;
    return sc_launch_run_lin(probe, 0, nullptr, 0) == 1;
}

extern "C" int sc_hk_run_supported(const sc_potential *pot, const sc_hk_consts *hk, const sc_overlap_consts *ovl) {
    if (!pot || !hk || !ovl) return 0;
    const bool sep = pot->kind == SC_POT_MORSE || pot->kind == SC_POT_HARMONIC_SEP || pot->kind == SC_POT_EPS_MORSE;
    if (sep && hk->diag && ovl->diag && pot->dim <= SC_SEP16_MAX_D) return 1;
    return run_lin_shape(pot, hk) ? 1 : 0;
}

static int run_whole_loop(const sc_potential *pot, const sc_state *st, const sc_hk_consts *hk, const sc_overlap_consts *ovl_t0,
                          const sc_nac_consts *nc, const double *vi, const double *probi, const double *nacq, double mc_norm,
                          double dt, int32_t nsteps, const double *mode_prop, double *partials, double *slots_out, double *elog, void *stream);

extern "C" int sc_hk_run(const sc_potential *pot, const sc_state *st, const sc_hk_consts *hk, const sc_overlap_consts *ovl_t0,
                         const sc_nac_consts *nc, const double *vi, const double *probi, const double *nacq, double mc_norm,
                         double dt, int32_t nsteps, double *partials, double *slots_out, double *elog, void *stream) {
    return run_whole_loop(pot, st, hk, ovl_t0, nc, vi, probi, nacq, mc_norm, dt, nsteps, nullptr, partials, slots_out, elog, stream);
}

extern "C" int sc_hk_run_modal_supported(const sc_potential *pot, const sc_hk_consts *hk, const sc_overlap_consts *ovl) {
    if (!pot || !hk || !ovl) return 0;
    return run_lin_shape(pot, hk, true) ? 1 : 0;
}

extern "C" int sc_hk_run_modal(const sc_potential *pot, const sc_state *st, const sc_hk_consts *hk, const sc_overlap_consts *ovl_t0,
                               const sc_nac_consts *nc, const double *vi, const double *probi, const double *nacq, double mc_norm,
                               double dt, int32_t nsteps, const double *mode_prop, double *partials, double *slots_out, double *elog,
                               void *stream) {
    if (!mode_prop) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_run_modal: null mode_prop");
    if (!pot || !hk || !ovl_t0 || !sc_hk_run_modal_supported(pot, hk, ovl_t0))
        return sc_fail(SC_ERR_UNSUPPORTED, "sc_hk_run_modal: needs a constant dense Hessian and dense real prefactor constants at an "
                       "instantiated shape D <= 16 (use sc_hk_run)");
    return run_whole_loop(pot, st, hk, ovl_t0, nc, vi, probi, nacq, mc_norm, dt, nsteps, mode_prop, partials, slots_out, elog, stream);
}

static int run_whole_loop(const sc_potential *pot, const sc_state *st, const sc_hk_consts *hk, const sc_overlap_consts *ovl_t0,
                          const sc_nac_consts *nc, const double *vi, const double *probi, const double *nacq, double mc_norm,
                          double dt, int32_t nsteps, const double *mode_prop, double *partials, double *slots_out, double *elog, void *stream) {
    if (!pot || !st || !hk || !ovl_t0 || !vi || !probi || !partials || !slots_out || !elog)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_run: null argument");
    if (nc && !nacq) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_run: nac constants without nacq");
    if (!mode_prop && !sc_hk_run_supported(pot, hk, ovl_t0))
        return sc_fail(SC_ERR_UNSUPPORTED, "sc_hk_run: needs a separable potential with diagonal width matrices and D <= %d, or a "
                       "constant dense Hessian with its step matrix (sc_potential.lin_prop) at an instantiated shape D <= 16 "
                       "(use the step-by-step entry points)", SC_SEP16_MAX_D);
    const bool lin = pot->kind == SC_POT_HARMONIC_DENSE;
    if (lin && !mode_prop && pot->lin_dt != dt)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_run: the step matrix was built for dt = %g, the call asks for %g", pot->lin_dt, dt);
    if (pot->dim != st->dim || hk->dim != st->dim || ovl_t0->dim != st->dim)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_run: dimension mismatch");
    if (int rq = sc_require_rowmajor(st, "sc_hk_run")) return rq;
    if (st->n <= 0 || nsteps <= 0) return SC_OK;
    hipStream_t s = (hipStream_t)stream;
    RunArgs a;
    a.step = StepArgs{*pot, *st, *hk, dt, 0, nullptr, 0};
    a.oc = *ovl_t0; a.has_nac = nc != nullptr;
    if (nc) a.nc = *nc; else a.nc = sc_nac_consts{};
    a.vi = vi; a.probi = probi; a.nacq = nacq; a.mc_norm = mc_norm; a.nsteps = nsteps; a.partials = partials;
    a.slots = sc_hk_run_slots(st->n, st->dim);
    a.mode_prop = mode_prop;
    if (hipMemsetAsync(partials, 0, sizeof(double) * 5 * (size_t)a.slots * (size_t)nsteps, s) != hipSuccess)
        return sc_check_launch("sc_hk_run (partials)");
    const int grid = a.slots / 4, D = st->dim;
    if (lin) {
        int rc = sc_launch_run_lin(a, grid, s, 1);
        if (rc < 0) return rc;
        if (rc == 0) return sc_fail(SC_ERR_UNSUPPORTED, "sc_hk_run: shape D=%d d'=%d not instantiated", D, hk->dprime);
        hipLaunchKernelGGL(hk_run_reduce_kernel, dim3(nsteps), dim3(256), 0, s, partials, a.slots, (double)st->n, slots_out);
        hipLaunchKernelGGL(hk_run_guard_kernel, dim3(1), dim3(64), 0, s, slots_out, nsteps, elog);
        return sc_check_launch("sc_hk_run (reduction)");
    }
#define SC_RUN_K(DP_, KIND_) hipLaunchKernelGGL((hk_run_sep16_kernel<DP_, KIND_>), dim3(grid), dim3(256), 0, s, a)
#define SC_RUN(DP_)                                                                     \
    do {                                                                                \
        if (pot->kind == SC_POT_MORSE) SC_RUN_K(DP_, SC_POT_MORSE);                     \
        else if (pot->kind == SC_POT_HARMONIC_SEP) SC_RUN_K(DP_, SC_POT_HARMONIC_SEP);  \
        else SC_RUN_K(DP_, SC_POT_EPS_MORSE);                                           \
    } while (0)
    if (D <= 4) SC_RUN(4);
    else if (D <= 8) SC_RUN(8);
    else SC_RUN(12);
#undef SC_RUN
#undef SC_RUN_K
    int rc = sc_check_launch("sc_hk_run (fused steps)");
    if (rc) return rc;
    hipLaunchKernelGGL(hk_run_reduce_kernel, dim3(nsteps), dim3(256), 0, s, partials, a.slots, (double)st->n, slots_out);
    hipLaunchKernelGGL(hk_run_guard_kernel, dim3(1), dim3(64), 0, s, slots_out, nsteps, elog);
    return sc_check_launch("sc_hk_run (reduction)");
}
