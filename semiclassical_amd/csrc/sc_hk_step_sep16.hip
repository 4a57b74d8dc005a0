// Fused Herman-Kluk step for SEPARABLE potentials with diagonal width matrices and small D: FOUR trajectories per
// wavefront, one per 16-lane DPP row (the layout of sc_wm_small.hip / sc_hk_step_lin.hip).
//
// Lane a < D of a row holds mode a and ROW a of the four monodromy blocks of its trajectory in registers (DP = D rounded up
// to a multiple of four is a template parameter: every register index is static).  With a diagonal Hessian the RK4 step of
// the monodromy rows is the 2 x 2 propagator P_a of the row (see sc_hk_step_sd.hip, hk_modes_kernel): the whole step is
// lane-local, no cross-lane traffic at all.  Then the diagonal-width prefactor row (reference propagators.py:969-986),
// the determinant by elimination in a FIXED pivot order -- the pivot row is a static lane, every update a fused broadcast
// multiply-add (sc_row16.h: det_rows_fixed_order; round 2 searched the pivot lane and fetched its row with ds_bpermute) --
// and the branch tracker.  A pivot more than 16 x below the best candidate flags the trajectory for the fully pivoted
// fix-up launch of sc_hk_step (never for the diagonal monodromy blocks a separable potential produces from M(0) = 1).  hk_step_w16_kernel spends a whole wavefront on a trajectory; at D = 5 that is 25 of 256 element slots.
// Workloads: BASELINE.json configs[0] (5 modes) and every separable / diagonal-width problem with D <= 12.
#include "sc_common.h"
#include "sc_row16.h"

// SC_SEP16_MAX_D (sc_common.h) = 12: 13 <= D <= 16 stay with hk_step_w16_kernel (measured in round 2 at n = 1e5, step launch
// in ms, packed / one wavefront per trajectory: D = 3: 0.07 / 0.24, 5: 0.12 / 0.29, 8: 0.18 / 0.35, 12: 0.37 / 0.50,
// 14: 0.58 / 0.58, 16: 0.84 / 0.63)
#ifndef SC_SEP16_FORCE_FIXUP
#define SC_SEP16_FORCE_FIXUP 0   // 1: variant library that hands every determinant to the pivoted fix-up launch
#endif

namespace {

typedef unsigned int sc_v2u __attribute__((ext_vector_type(2)));

// KIND = potential family (SC_POT_MORSE, SC_POT_HARMONIC_SEP, SC_POT_EPS_MORSE) as a template parameter: one formula per
// instantiation (with a run-time switch all three, each with its exp, are live at once: 216 VGPRs at DP = 4)
template <int DP, bool STEP, int KIND>
__global__ __launch_bounds__(256, 2) void hk_step_sep16_kernel(StepArgs A) {
    __shared__ double red[16];
    const int D = A.st.dim, DD = D * D, tid = threadIdx.x, r = tid & 15, grp = tid >> 4;
    const double dt = A.dt, hh = 0.5 * dt, h6 = dt / 6.0;
    const bool mine = r < D;
    const double sta = mine ? A.hk.st[r] : 1.0, ista = 1.0 / sta;
    const double im = mine ? A.pot.inv_mass[r] : 0.0, c0 = mine ? A.pot.par0[r] : 0.0;
    const double c1 = (mine && A.pot.par1) ? A.pot.par1[r] : 0.0;
    kptr ksi = (kptr)A.hk.si;
    double esum = 0.0;
    const int64_t n = A.st.n, stride = (int64_t)gridDim.x * 16;
    // Addressing: raw buffer instructions on two resources per workgroup pass -- the monodromy blocks and (q, p) of the 16
    // trajectories t0 .. t0 + 15 -- with ONE 32-bit VGPR offset per thread (row a of its trajectory), the column as the
    // instruction's immediate offset, the block as scalar offset.  The resources end behind the last existing trajectory and
    // lanes without a row carry an offset beyond them: their loads return 0, their stores are dropped.  (Pointer
    // arithmetic made hipcc keep a 64-bit address per element: 166 VGPRs at DP = 4.)
    constexpr unsigned OOB = 0x7fffffffu;
    const unsigned vo_m = mine ? 8u * (unsigned)(grp * 4 * DD + r * D) : OOB, vo_q = mine ? 8u * (unsigned)(grp * 2 * D + r) : OOB;
    auto ldg = [](__amdgpu_buffer_rsrc_t rs, unsigned vo, int so) {
        const sc_v2u v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)vo, so, 0);
        return __hiloint2double((int)v.y, (int)v.x);
    };
    auto stg = [](double x, __amdgpu_buffer_rsrc_t rs, unsigned vo, int so) {
        sc_v2u v;
        v.x = (unsigned)__double2loint(x); v.y = (unsigned)__double2hiint(x);
        __builtin_amdgcn_raw_buffer_store_b64(v, rs, (int)vo, so, 0);
    };
    for (int64_t t0 = (int64_t)blockIdx.x * 16; t0 < n; t0 += stride) {
        const int have = (int)(n - t0 < 16 ? n - t0 : 16);
        const bool active = grp < have;
        const int64_t tr = t0 + grp;
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(A.st.mono + t0 * 4 * (int64_t)DD, 0, have * 32 * DD, 0x00020000);
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(A.st.qp + t0 * 2 * D, 0, have * 16 * D, 0x00020000);
        // rows of the four blocks: requested first, the mode's RK4 below runs under the loads
        double mqq[DP], mqp[DP], mpq[DP], mpp[DP];
#pragma unroll
        for (int b = 0; b < DP; ++b) {
            const unsigned vo = b < D ? vo_m + 8u * b : OOB;
            mqq[b] = ldg(rm, vo, 0); mqp[b] = ldg(rm, vo, 8 * DD); mpq[b] = ldg(rm, vo, 16 * DD); mpp[b] = ldg(rm, vo, 24 * DD);
        }
        if (STEP) {
            // ---- RK4 of (q_a, p_a) with the reference's stage formula, action, <T+V> and the row propagator P_a
            double red5[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, p11 = 1.0, p12 = 0.0, p21 = 0.0, p22 = 1.0;
            if (mine) {
                const double q = ldg(rq, vo_q, 0), p = ldg(rq, vo_q, 8 * D);
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
                stg(q + h6 * (kq1 + 2.0 * kq2 + 2.0 * kq3 + kq4), rq, vo_q, 0);
                stg(p + h6 * (kp1 + 2.0 * kp2 + 2.0 * kp3 + kp4), rq, vo_q, 8 * D);
                p11 = 1.0; p21 = 0.0; p12 = 0.0; p22 = 1.0;
                rk4_pair(p11, p21, im, h1, h2, h3, h4, dt);       // (u, v) = (1, 0) -> first column of P_a
                rk4_pair(p12, p22, im, h1, h2, h3, h4, dt);       // (0, 1) -> second column
            }
            {
                double one = 1.0, s5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
                asm volatile("" : "+v"(one));
                dpp_guard(red5);
                sfor<0, DP>([&](auto kc) {            // sums over the lanes k < DP (lanes beyond D hold zeros)
#pragma unroll
                    for (int i = 0; i < 5; ++i) fmac_bc<decltype(kc)::value>(s5[i], red5[i], one);
                });
#pragma unroll
                for (int i = 0; i < 5; ++i) red5[i] = s5[i];
            }
            if (active && r == 0) {
                A.st.act[tr] += h6 * (red5[0] + 2.0 * red5[1] + 2.0 * red5[2] + red5[3]);
                esum += red5[4];
            }
            // ---- (Mqq, Mpq)' = P_a (Mqq, Mpq), (Mqp, Mpp)' = P_a (Mqp, Mpp)
#pragma unroll
            for (int b = 0; b < DP; ++b) {
                const double nqq = fma(p12, mpq[b], p11 * mqq[b]), npq = fma(p22, mpq[b], p21 * mqq[b]);
                const double nqp = fma(p12, mpp[b], p11 * mqp[b]), npp = fma(p22, mpp[b], p21 * mqp[b]);
                mqq[b] = nqq; mpq[b] = npq; mqp[b] = nqp; mpp[b] = npp;
                const unsigned vo = b < D ? vo_m + 8u * b : OOB;
                stg(nqq, rm, vo, 0); stg(nqp, rm, vo, 8 * DD); stg(npq, rm, vo, 16 * DD); stg(npp, rm, vo, 24 * DD);
            }
        }
        // ---- prefactor row a (rows and columns beyond D: identity, lanes beyond DP: no row)
        cplx mat[DP];
#pragma unroll
        for (int b = 0; b < DP; ++b) {
            const double sib = b < D ? ksi[b] : 1.0, isib = 1.0 / sib;
            mat[b] = (mine && b < D) ? c_make(0.5 * (sta * isib * mqq[b] + ista * sib * mpp[b]),
                                              0.5 * (-SC_HBAR * sta * sib * mqp[b] + (1.0 / SC_HBAR) * ista * isib * mpq[b]))
                                     : c_make(r == b ? 1.0 : 0.0, 0.0);
        }
        int weak = SC_SEP16_FORCE_FIXUP;
        const cplx det = det_rows_fixed_order<DP>(mat, r, weak);
        if (active && r == 0 && weak && A.st.flags) {
            A.st.flags[tr] = 1;                       // c2 / sgn are left to the fully pivoted fix-up launch
            atomicAdd(&A.st.flags[n], 1);
        } else if (active && r == 0) {
            cplx *c2 = (cplx *)A.st.c2;
            if (STEP) {
                const cplx prev = c2[tr];
                if (prev.x < 0.0 && det.x < 0.0 && prev.y * det.y < 0.0) A.st.sgn[tr] = -A.st.sgn[tr];
            } else {
                A.st.sgn[tr] = 1.0;
            }
            c2[tr] = det;
        }
    }
    if (r == 0) red[grp] = esum;
    __syncthreads();
    if (tid == 0 && A.epart && STEP) {
        double s = 0.0;
        for (int g = 0; g < 16; ++g) s += red[g];
        A.epart[blockIdx.x] = s;
    }
    // the energy guard adds sc_step_grid(n, D) = min(n, 4096) partials: workgroup 0 zeroes the ones no workgroup owns
    // (instead of a memset launch in front of every step)
    if (blockIdx.x == 0 && A.epart && STEP) {
        const int entries = (int)(n < 4096 ? n : 4096);
        for (int i = gridDim.x + tid; i < entries; i += 256) A.epart[i] = 0.0;
    }
}

}  // namespace

// launches for D <= SC_SEP16_MAX_D and returns 1; 0 if the dimension is left to hk_step_w16_kernel; < 0 on error.
// `grid_entries` = sc_step_grid(n, D): the energy partials the guard adds up (zeroed here, the kernel writes the first ones)
int sc_launch_step_sep16(const StepArgs &a, int grid_entries, hipStream_t s) {
    const int D = a.st.dim;
    if (D > SC_SEP16_MAX_D) return 0;
    const int64_t groups = (a.st.n + 15) / 16;
    const int wg = (int)(groups < 4096 ? groups : 4096);
    if (wg > grid_entries) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_step: %d energy partials, %d workgroups", grid_entries, wg);
    const bool step = (a.mode & 0xff) == 0;
    if (grid_entries != (int)(a.st.n < 4096 ? a.st.n : 4096))
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_step: %d energy partials, the kernel zeroes min(n, 4096)", grid_entries);
#define SC_SEP16_K(DP_, KIND_)                                                                                      \
    do {                                                                                                            \
        if (step) hipLaunchKernelGGL((hk_step_sep16_kernel<DP_, true, KIND_>), dim3(wg), dim3(256), 0, s, a);       \
        else hipLaunchKernelGGL((hk_step_sep16_kernel<DP_, false, SC_POT_HARMONIC_SEP>), dim3(wg), dim3(256), 0, s, a); \
    } while (0)
#define SC_SEP16(DP_)                                                                                               \
    do {                                                                                                            \
        if (a.pot.kind == SC_POT_MORSE) SC_SEP16_K(DP_, SC_POT_MORSE);                                              \
        else if (a.pot.kind == SC_POT_HARMONIC_SEP) SC_SEP16_K(DP_, SC_POT_HARMONIC_SEP);                           \
        else SC_SEP16_K(DP_, SC_POT_EPS_MORSE);                                                                     \
    } while (0)
    if (D <= 4) SC_SEP16(4);
    else if (D <= 8) SC_SEP16(8);
    else SC_SEP16(12);
#undef SC_SEP16_K
#undef SC_SEP16
    const int rc = sc_check_launch("sc_hk_step (four trajectories per wavefront)");
    return rc == SC_OK ? 1 : rc;
}
