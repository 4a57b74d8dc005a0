// Shared device helpers of the gfx950 trajectory engine (wave64, fp64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "semiclassical_hip.h"

#define SC_HBAR 1.0   // reference semiclassical/units.py:8

typedef double2 cplx;   // (re, im), same memory layout as torch.complex128

__device__ __forceinline__ cplx c_make(double re, double im) { return make_double2(re, im); }
__device__ __forceinline__ cplx c_add(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx c_sub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx c_mul(cplx a, cplx b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ cplx c_scale(cplx a, double s) { return make_double2(a.x * s, a.y * s); }
__device__ __forceinline__ cplx c_conj(cplx a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ double c_abs2(cplx a) { return a.x * a.x + a.y * a.y; }
__device__ __forceinline__ cplx c_inv(cplx a) {
    double s = 1.0 / c_abs2(a);
    return make_double2(a.x * s, -a.y * s);
}
// a - l*u
__device__ __forceinline__ cplx c_fnma(cplx l, cplx u, cplx a) {
    a.x = fma(-l.x, u.x, a.x); a.x = fma(l.y, u.y, a.x);
    a.y = fma(-l.x, u.y, a.y); a.y = fma(-l.y, u.x, a.y);
    return a;
}
// a + l*u
__device__ __forceinline__ cplx c_fma(cplx l, cplx u, cplx a) {
    a.x = fma(l.x, u.x, a.x); a.x = fma(-l.y, u.y, a.x);
    a.y = fma(l.x, u.y, a.y); a.y = fma(l.y, u.x, a.y);
    return a;
}
// principal square root (branch cut on the negative real axis, sign of Im follows Im z), as torch.sqrt
__device__ __forceinline__ cplx c_sqrt(cplx z) {
    double r = hypot(z.x, z.y);
    if (r == 0.0) return make_double2(0.0, z.y);
    if (z.x >= 0.0) {
        double t = sqrt(0.5 * (r + z.x));
        return make_double2(t, z.y / (2.0 * t));
    }
    double t = sqrt(0.5 * (r - z.x));
    return make_double2(fabs(z.y) / (2.0 * t), copysign(t, z.y));
}
__device__ __forceinline__ cplx c_exp(cplx z) {
    double s, c;
    sincos(z.y, &s, &c);
    double e = exp(z.x);
    return make_double2(e * c, e * s);
}

// Sum over the wavefront, result in every lane.  DPP rotations inside each 16-lane row (64-bit values move as two
// 32-bit DPP moves), then one v_readlane pair per row: no LDS traffic (`__shfl_xor` is a ds_bpermute, ~100 cycles
// per dependent step, 12 of them for a double).  Fixed order => deterministic.
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov_f64<0x128>(v);      // row_ror:8
    v += dpp_mov_f64<0x124>(v);      // row_ror:4
    v += dpp_mov_f64<0x122>(v);      // row_ror:2
    v += dpp_mov_f64<0x121>(v);      // row_ror:1
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return (r0 + r1) + (r2 + r3);
}

// Wave-uniform maximum of a 32-bit integer over the wavefront without LDS traffic: DPP row rotations inside each
// 16-lane row, then one v_readlane per row.  (`__shfl_xor` is a ds_bpermute: ~100 cycles per dependent step.)
__device__ __forceinline__ int wave_max_i32(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false));      // row_ror:8
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xF, 0xF, false));      // row_ror:4
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xF, 0xF, false));      // row_ror:2
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xF, 0xF, false));      // row_ror:1
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

// Partial-pivoting search over one candidate per lane: `mag` = |a|^2 of this lane's candidate row `row` (or any
// value with valid = false).  Returns the row of (nearly) the largest magnitude: the comparison key keeps the sign,
// the exponent and 14 mantissa bits of |a|^2, ties go to the lower lane.  -1 if no lane is valid.
__device__ __forceinline__ int wave_pivot_row(double mag, int row, bool valid) {
    const int lane = threadIdx.x & 63;
    const int key = valid ? ((__double2hiint(mag) & ~63) | (63 - lane)) : -1;
    const int best = wave_max_i32(key);
    if (best < 0) return -1;
    return __builtin_amdgcn_readlane(row, __builtin_amdgcn_readfirstlane(63 - (best & 63)));
}

// Sum NV values over the whole workgroup; every thread receives the totals.
// `red` is LDS scratch of at least NV * (blockDim.x / 64) doubles.  Deterministic order.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    if (nw == 1) return;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += red[w * NV + i];
        v[i] = s;
    }
}

// offset of element (a, b) of block p inside one trajectory's monodromy storage (include/semiclassical_hip.h)
__host__ __device__ __forceinline__ int64_t sc_mono_offset(int layout, int D, int p, int a, int b) {
    if (layout == SC_MONO_ROWMAJOR) return ((int64_t)p * D + a) * D + b;
    const int ra = a >> 4, rb = b >> 4;
    const int nra = D - 16 * ra < 16 ? D - 16 * ra : 16, ncb = D - 16 * rb < 16 ? D - 16 * rb : 16;
    // inside a tile: (Mqq, Mqp) element by element, then (Mpq, Mpp) element by element (a thread of the fast kernel moves the two
    // planes of a pair with ONE 16-byte access; round 4 -- the four planes used to follow one another)
    return 4 * ((int64_t)16 * ra * D + 16 * nra * rb) + (int64_t)(p >> 1) * 2 * nra * ncb + 2 * ((a & 15) * ncb + (b & 15)) + (p & 1);
}

// ---- shared by the step kernels ----
struct StepArgs {
    sc_potential pot;
    sc_state st;
    sc_hk_consts hk;
    double dt;
    int mode;
    double *epart;
    int npart;          // entries of epart the energy guard adds up (sc_step_grid); kernels with fewer workgroups clear the rest
};

// V, dV/dx, d2V/dx2 of one mode of a separable potential
__device__ __forceinline__ void sep_eval(int kind, double c0, double c1, double x, double &v, double &g, double &h) {
    if (kind == SC_POT_MORSE) {                    // c0 = a, c1 = De
        double e = exp(-c0 * x);
        double om = 1.0 - e;
        v = c1 * om * om;
        g = 2.0 * c0 * c1 * e * om;
        h = 2.0 * c0 * c0 * c1 * e * (2.0 * e - 1.0);
    } else if (kind == SC_POT_HARMONIC_SEP) {      // c0 = omega^2
        v = 0.5 * c0 * x * x;
        g = c0 * x;
        h = c0;
    } else {                                       // SC_POT_EPS_MORSE: c0 = eps, c1 = b
        double e1 = exp(-c1 * x), e2 = exp(-2.0 * c1 * x);
        double om = 1.0 - e1;
        v = c0 / (2.0 * c1 * c1) * om * om + (1.0 - c0) * 0.5 * x * x;
        g = c0 / c1 * (e1 - e2) + (1.0 - c0) * x;
        h = c0 * (2.0 * e2 - e1) + (1.0 - c0);
    }
}

// RK4 of the pair (u, v) with du/dt = v/m, dv/dt = -h(t) u  (diagonal Hessian)
__device__ __forceinline__ void rk4_pair(double &u, double &v, double im, double h1, double h2, double h3, double h4,
                                         double dt) {
    const double hh = 0.5 * dt, h6 = dt / 6.0;
    double k1u = v * im, k1v = -h1 * u;
    double u2 = u + hh * k1u, v2 = v + hh * k1v;
    double k2u = v2 * im, k2v = -h2 * u2;
    double u3 = u + hh * k2u, v3 = v + hh * k2v;
    double k3u = v3 * im, k3v = -h3 * u3;
    double u4 = u + dt * k3u, v4 = v + dt * k3v;
    double k4u = v4 * im, k4v = -h4 * u4;
    u = u + h6 * (k1u + 2.0 * k2u + 2.0 * k3u + k4u);
    v = v + h6 * (k1v + 2.0 * k2v + 2.0 * k3v + k4v);
}

#define SC_SEP16_MAX_D 12    // D <= 12: hk_step_sep16_kernel (four trajectories per wavefront); 13 .. 16: hk_step_w16_kernel
int sc_launch_step_sep16(const StepArgs &a, int grid_entries, hipStream_t s);   // sc_hk_step_sep16.hip: 1 launched, 0 not taken
int sc_launch_step_sd_multi(const StepArgs &a, const sc_multi_scratch &ms, hipStream_t s);    // sc_hk_step_sd.hip: two steps per visit
int sc_launch_step_lin(const StepArgs &a, int grid, hipStream_t s);   // sc_hk_step_lin.hip: 1 launched, 0 shape not built

// host-side error plumbing (sc_api.hip)
int sc_fail(int code, const char *fmt, ...);
int sc_check_launch(const char *what);
int sc_require_rowmajor(const sc_state *st, const char *who);     // sc_layout.hip
