// "One trajectory per 16-lane DPP row" building blocks shared by the register-resident small-matrix kernels
// (sc_wm_small.hip, sc_hk_step_lin.hip): static loops with one basic block per iteration, 64-bit DPP row broadcasts,
// row reductions, ds_bpermute with a data-dependent lane, and Gauss-Jordan elimination with the pivot chosen among LANES.
#pragma once
#include "sc_common.h"

#include <type_traits>

namespace {

typedef const __attribute__((address_space(4))) double *kptr;   // uniform read-only data: scalar loads

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

// A uniform branch the compiler cannot fold.  The body of a trajectory is several thousand independent multiply-adds
// and cross-lane moves; as ONE basic block the instruction selector's list scheduler issues every move of a phase
// first and the register allocator spills them (measured: 980 spilled VGPRs).  Wrapping every phase and every
// iteration of the unrolled product loops in `if (opaque_true())` makes each its own basic block: nothing is
// scheduled across, live ranges stay those of the source order.  Cost: three scalar instructions per block.
__device__ __forceinline__ bool opaque_true() {
    int one = 1;
    asm volatile("" : "+s"(one));
    return one != 0;
}
#define WM_BLOCK if (opaque_true())

// sfor with every iteration in a basic block of its own
template <int I, int N, class F>
__device__ __forceinline__ void sfor_bb(F &&f) {
    if constexpr (I < N) {
        WM_BLOCK { f(std::integral_constant<int, I>{}); }
        sfor_bb<I + 1, N>(f);
    }
}

// value of lane K of this lane's 16-lane row (v_mov_b64_dpp row_newbcast:K)
template <int K>
__device__ __forceinline__ double bc(double v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xF, 0xF, true);
}

// sum over the 16 lanes of the row, result in every lane of the row; fixed order
__device__ __forceinline__ double row_sum(double v) {
    v += dpp_mov_f64<0x128>(v);
    v += dpp_mov_f64<0x124>(v);
    v += dpp_mov_f64<0x122>(v);
    v += dpp_mov_f64<0x121>(v);
    return v;
}
__device__ __forceinline__ int row_max(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xF, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xF, 0xF, false));
    return v;
}
// value of `v` in the lane whose byte index (4 * lane) is `addr`
__device__ __forceinline__ double perm(int addr, double v) {
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ cplx perm(int addr, cplx v) { return c_make(perm(addr, v.x), perm(addr, v.y)); }

// LDS traffic of ONE wavefront executes in issue order: a fence for the compiler is all a write -> read hand-over
// between lanes of the same wavefront needs
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Gauss-Jordan elimination of the N x N complex system held one ROW per lane (lane i: a[0..N) and NR right-hand
// sides b[0..NR)); rows >= N must be zero and enter with used = true.  Partial pivoting over the unused LANES for
// column k.  Rows are not normalised while eliminating: a lane subtracts m = a_ik / pivot times the pivot row, the
// pivot lane itself takes m = 0 (one select per step instead of one per element), and at the end every lane divides
// its right-hand sides by its own pivot.  On exit the lane that was the pivot of step k (`myk` = k) holds row k of
// the solution in b; `src` of lane k is the byte address (ds_bpermute) of that lane; det = determinant of the matrix.
template <int N, int NR>
__device__ __forceinline__ void gauss_jordan_rows(cplx (&a)[N], cplx (&b)[NR], bool used, int r, int rowbase, int &myk,
                                                  int &src, cplx &det) {
    det = c_make(1.0, 0.0);
    int parity = 0;
    bool singular = false;
    cplx myinv = c_make(0.0, 0.0);
    myk = r;
    src = (rowbase | r) << 2;
    sfor_bb<0, N>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        // pivot lane: largest |a_ik|^2 among the unused lanes (upper 28 bits of the magnitude, ties to the lower lane)
        const int key = used ? -1 : ((__double2hiint(c_abs2(a[k])) & ~15) | (15 - r));
        const int best = row_max(key);
        const int p = 15 - (best & 15);
        const int addr = (rowbase | p) << 2;
        const unsigned long long unused = __ballot(!used);
        parity ^= __popcll((unused >> rowbase) & ((1ull << p) - 1ull)) & 1;
        const cplx piv = perm(addr, a[k]);
        singular = singular || (piv.x == 0.0 && piv.y == 0.0);
        det = c_mul(det, piv);
        const cplx inv = c_inv(piv);
        const bool me = r == p;
        const cplx f = c_mul(a[k], inv);
        const cplx m = c_make(me ? 0.0 : f.x, me ? 0.0 : f.y);
        sfor<k + 1, N>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            a[j] = c_fnma(m, perm(addr, a[j]), a[j]);
        });
        sfor<0, NR>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            b[j] = c_fnma(m, perm(addr, b[j]), b[j]);
        });
        if (me) { used = true; myk = k; myinv = inv; }
        if (r == k) src = addr;
    });
#pragma unroll
    for (int j = 0; j < NR; ++j) b[j] = c_mul(b[j], myinv);
    if (parity) det = c_make(-det.x, -det.y);
    if (singular) det = c_make(0.0, 0.0);
}

}  // namespace
