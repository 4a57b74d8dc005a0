// Shared declarations of the Walton-Manolopoulos kernels (sc_wm.hip: general LDS / global-scratch kernel and the C-ABI
// entry points; sc_wm_small.hip: register-resident kernel for small matrices).
#pragma once
#include "sc_common.h"

struct WmArgs {
    sc_state st;
    sc_wm_consts wc;
    const double *zi, *probi;
    double mc_norm;
    int track;          // 0: use the stored branch signs, 1: track against the previous determinants, 2: initialise
    int has_nac;
    int stage_consts;   // general kernel: copy the D x D constants into LDS once per workgroup (when they fit)
    double *cq_out, *kq_out, *partials;
    double *scratch;    // general kernel, large D: per-workgroup matrix storage in global memory (NULL: LDS)
    size_t scratch_stride;   // bytes per workgroup
    int npartials;      // slots of `partials` the caller sums (sc_wm_grid); the kernels fill all of them
    const int32_t *only_flagged;   // general kernel: NULL, or [n + 1] flags -- process trajectory i only if only_flagged[i] != 0
};

// branch tracker of sqrt(z(t)), reference propagators.py:1006-1052; returns the sign to use now
__device__ __forceinline__ double wm_track_sign(int track, cplx z, cplx *prev, double *sgn) {
    double s = *sgn;
    if (track == 2) { s = 1.0; *sgn = s; *prev = z; }
    else if (track == 1) {
        const cplx z1 = *prev;
        if (z1.x < 0.0 && z.x < 0.0 && z1.y * z.y < 0.0) s = -s;
        *sgn = s; *prev = z;
    }
    return s;
}

// doubles per trajectory the register kernel hands to its tail kernel through wc->scratch: detA, detM, ex, (eps, dq.Cqq.dq),
// (pi_q.dq, -), nacQ, nacq, nacqQ
#define WM_TAIL_FIELDS 16

// sc_wm_small.hip: returns 1 and launches if (D, d') has a register-resident instantiation, 0 if not, < 0 on error
int sc_wm_launch_small(const WmArgs &a, int grid, hipStream_t s);
