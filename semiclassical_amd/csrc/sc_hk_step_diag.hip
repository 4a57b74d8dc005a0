// Structure-exploiting Herman-Kluk step ("separable shortcut"): for a SEPARABLE potential
// (diagonal Hessian), DIAGONAL width matrices and monodromy blocks that are diagonal at the start (M(0) = 1, the
// state initial_conditions() creates) the four monodromy blocks stay diagonal for all times, the prefactor matrix
//   mat_aa = 1/2[ st_a/si_a Mqq_aa + si_a/st_a Mpp_aa - i hbar st_a si_a Mqp_aa + i/hbar Mpq_aa/(st_a si_a) ]
// is diagonal and its determinant is the product of its diagonal.  The state shrinks from 4 D^2 to 4 D doubles per
// trajectory; the results are those of the dense kernels up to the rounding of the determinant.
// It is OPT-IN on the host (HermanKlukPropagator(..., exploit_separability=True)) and is reported separately from the
// dense-state kernel in bench.py, with its own byte model ((12 D + 7) * 8 bytes per trajectory step).
//
// One wavefront per trajectory, lane = mode (modes beyond 64 in further passes).  Same arithmetic as the dense fast
// path: RK4 of (q_a, p_a) with the reference's stage formula (propagators.py:86-119, 313-383), the 2x2 RK4
// propagator P_a of the monodromy rows, the action, <T+V> at the k4 stage, then the branch tracker (:1006-1052).
#include "sc_common.h"

namespace {

__device__ __forceinline__ cplx wave_prod(cplx z) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const cplx o = c_make(__shfl_xor(z.x, off, 64), __shfl_xor(z.y, off, 64));
        z = c_mul(z, o);
    }
    return z;
}

template <bool STEP>
__global__ __launch_bounds__(256) void hk_diag_step_kernel(StepArgs A, double *mdiag) {
    const int D = A.st.dim, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double dt = A.dt, hh = 0.5 * dt, h6 = dt / 6.0;
    __shared__ double wsum[4];
    double esum = 0.0;
    for (int64_t tr = (int64_t)blockIdx.x * 4 + wave; tr < A.st.n; tr += (int64_t)gridDim.x * 4) {
        double *qp = A.st.qp + tr * 2 * D;
        double *md = mdiag + tr * 4 * (int64_t)D;
        double red5[5] = {0, 0, 0, 0, 0};
        cplx det = c_make(1.0, 0.0);
        for (int a = lane; a < ((D + 63) & ~63); a += 64) {
            cplx mat = c_make(1.0, 0.0);
            if (a < D) {
                double mqq = md[a], mqp = md[D + a], mpq = md[2 * D + a], mpp = md[3 * D + a];
                if (STEP) {
                    const double q = qp[a], p = qp[D + a], im = A.pot.inv_mass[a];
                    const double c0 = A.pot.par0[a], c1 = A.pot.par1 ? A.pot.par1[a] : 0.0;
                    double v, g, h1, h2, h3, h4;
                    sep_eval(A.pot.kind, c0, c1, q, v, g, h1);
                    const double kq1 = p * im, kp1 = -g;
                    red5[0] += 0.5 * p * p * im - v;
                    const double q2 = q + hh * kq1, p2 = p + hh * kp1;
                    sep_eval(A.pot.kind, c0, c1, q2, v, g, h2);
                    const double kq2 = p2 * im, kp2 = -g;
                    red5[1] += 0.5 * p2 * p2 * im - v;
                    const double q3 = q + hh * kq2, p3 = p + hh * kp2;
                    sep_eval(A.pot.kind, c0, c1, q3, v, g, h3);
                    const double kq3 = p3 * im, kp3 = -g;
                    red5[2] += 0.5 * p3 * p3 * im - v;
                    const double q4 = q + dt * kq3, p4 = p + dt * kp3;
                    sep_eval(A.pot.kind, c0, c1, q4, v, g, h4);
                    const double kq4 = p4 * im, kp4 = -g;
                    red5[3] += 0.5 * p4 * p4 * im - v;
                    red5[4] += 0.5 * p4 * p4 * im + v;
                    qp[a] = q + h6 * (kq1 + 2.0 * kq2 + 2.0 * kq3 + kq4);
                    qp[D + a] = p + h6 * (kp1 + 2.0 * kp2 + 2.0 * kp3 + kp4);
                    double u1 = 1.0, v1 = 0.0, u2 = 0.0, v2 = 1.0;       // P_a = [[u1, u2], [v1, v2]]
                    rk4_pair(u1, v1, im, h1, h2, h3, h4, dt);
                    rk4_pair(u2, v2, im, h1, h2, h3, h4, dt);
                    const double nqq = fma(u2, mpq, u1 * mqq), npq = fma(v2, mpq, v1 * mqq);
                    const double nqp = fma(u2, mpp, u1 * mqp), npp = fma(v2, mpp, v1 * mqp);
                    mqq = nqq; mpq = npq; mqp = nqp; mpp = npp;
                    md[a] = mqq; md[D + a] = mqp; md[2 * D + a] = mpq; md[3 * D + a] = mpp;
                }
                const double st = A.hk.st[a], si = A.hk.si[a], ist = 1.0 / st, isi = 1.0 / si;
                mat = c_make(0.5 * (st * isi * mqq + ist * si * mpp),
                             0.5 * (-SC_HBAR * st * si * mqp + (1.0 / SC_HBAR) * ist * isi * mpq));
            }
            det = c_mul(det, mat);
        }
        det = wave_prod(det);
        if (STEP) {
#pragma unroll
            for (int i = 0; i < 5; ++i) red5[i] = wave_sum(red5[i]);
        }
        if (lane == 0) {
            cplx *c2 = (cplx *)A.st.c2;
            if (STEP) {
                A.st.act[tr] += h6 * (red5[0] + 2.0 * red5[1] + 2.0 * red5[2] + red5[3]);
                esum += red5[4];
                const cplx prev = c2[tr];
                if (prev.x < 0.0 && det.x < 0.0 && prev.y * det.y < 0.0) A.st.sgn[tr] = -A.st.sgn[tr];
            } else {
                A.st.sgn[tr] = 1.0;
            }
            c2[tr] = det;
        }
    }
    if (lane == 0) wsum[wave] = esum;
    __syncthreads();
    if (threadIdx.x == 0 && A.epart) A.epart[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

}  // namespace

extern "C" int sc_hk_step_diag(const sc_potential *pot, const sc_state *st, const sc_hk_consts *hk, double *mono_diag,
                               double dt, int32_t mode, double *energy_partials, void *stream) {
    if (!pot || !st || !hk || !mono_diag) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_step_diag: null argument");
    const int D = st->dim;
    if (pot->dim != D || hk->dim != D) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_step_diag: dimension mismatch");
    if (D < 1) return sc_fail(SC_ERR_UNSUPPORTED, "sc_hk_step_diag: D=%d", D);
    if (!hk->diag || hk->dprime != D)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_step_diag: needs diagonal width matrices without zero modes");
    if (pot->kind != SC_POT_MORSE && pot->kind != SC_POT_HARMONIC_SEP && pot->kind != SC_POT_EPS_MORSE)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_hk_step_diag: potential kind %d is not separable", pot->kind);
    if (st->n <= 0) return SC_OK;
    StepArgs a{*pot, *st, *hk, dt, mode, energy_partials, sc_step_grid(st->n, st->dim)};
    const int grid = sc_step_grid(st->n, D);
    if ((mode & 0xff) == 0) hipLaunchKernelGGL(hk_diag_step_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, mono_diag);
    else hipLaunchKernelGGL(hk_diag_step_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, mono_diag);
    return sc_check_launch("sc_hk_step_diag");
}
