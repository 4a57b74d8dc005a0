// Shared device helpers of the gfx950 trajectory engine (wave64, fp64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>

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

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
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

// host-side error plumbing (sc_api.hip)
int sc_fail(int code, const char *fmt, ...);
int sc_check_launch(const char *what);
