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

// ---- broadcast FUSED into the multiply-add (round 3) ----
// v_fmac_f64 is the one 64-bit VALU operation that takes a DPP operand on gfx90a+/gfx950 (DP-ALU DPP, row_newbcast only):
//     acc += b[lane K of this lane's 16-lane row] * a          in ONE instruction, at the rate of a plain v_fmac_f64
// (tools/micro/dpp_fmac.hip, profiles/r3_dpp_fmac.txt: 1.7x the rate of v_mov_b64_dpp + v_fmac_f64, identical results).
// hipcc never forms it (its DPP combiner runs while the multiply-add is still the three-address VOP3 v_fma_f64), hence
// inline assembly.  `volatile`: the statements keep their source order -- the register pressure of a product loop is the
// one written down, not what a list scheduler makes of several thousand independent multiply-adds.
// HAZARD the compiler cannot see: a VGPR written by a VALU instruction must not be read as the DPP operand by one of the
// next two instructions.  The DPP operands of every loop below are arrays completed in an earlier phase; `dpp_guard(...)`
// separates a phase from the code that produced its operands, and tests/test_host.py::test_dpp_hazards scans the
// disassembly of the built library for any DPP read closer than that to a write of the same register.
// `dpp_guard(operands...)` first pins every operand (scalar, complex or array) into its registers -- an empty volatile asm
// with the value as in/out operand: the compiler has to have computed or copied it BEFORE that point -- and then waits.
__device__ __forceinline__ void dpp_pin(double &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void dpp_pin(cplx &x) { asm volatile("" : "+v"(x.x), "+v"(x.y)); }
template <class T, int N>
__device__ __forceinline__ void dpp_pin(T (&x)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) dpp_pin(x[i]);
}
template <class... T>
__device__ __forceinline__ void dpp_guard(T &...operands) {
    (dpp_pin(operands), ...);
    asm volatile("s_nop 1");
}

template <int K>
__device__ __forceinline__ void fmac_bc(double &acc, const double &b, const double &a) {      // acc += b[K] * a
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(a), "n"(K));
}
template <int K>
__device__ __forceinline__ void fnmac_bc(double &acc, const double &b, const double &a) {     // acc -= b[K] * a
    asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(a), "n"(K));
}
// c += w * x[K] and c -= w * x[K] (complex; x is taken from lane K, w is this lane's).  c and x must be DIFFERENT
// registers (an in-place update would read x as DPP operand right behind its own write: cfnma_inplace below).
template <int K>
__device__ __forceinline__ void cfma_bc(cplx &c, const cplx &x, const cplx &w) {
    asm volatile("v_fmac_f64_dpp %0, %2, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %3, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %3, -%5 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %2, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf"
                 : "+v"(c.x), "+v"(c.y) : "v"(x.x), "v"(x.y), "v"(w.x), "v"(w.y), "n"(K));
}
template <int K>
__device__ __forceinline__ void cfnma_bc(cplx &c, const cplx &x, const cplx &w) {
    asm volatile("v_fmac_f64_dpp %0, %2, -%4 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %3, -%4 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %3, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %2, -%5 row_newbcast:%6 row_mask:0xf bank_mask:0xf"
                 : "+v"(c.x), "+v"(c.y) : "v"(x.x), "v"(x.y), "v"(w.x), "v"(w.y), "n"(K));
}

// IN-PLACE elimination update  c_j -= w * c_j[K]  of N array elements (the pivot lane K's own w must be zero, so that
// what the other lanes read from it does not change under the update).  The four multiply-adds of every element are
// issued phase by phase over the group -- all "x -= X.x w.x", all "y -= X.y w.x", all "x += X.y w.y", all "y -= X.x w.y"
// -- so that a register is read as DPP operand no earlier than N - 1 instructions after it was written: with N >= 3 the
// two wait states the hardware wants lie in between without a single s_nop.  Groups of one and two carry the s_nop.
#define SC_IP_A(c) "v_fmac_f64_dpp %[" #c "x], %[" #c "x], -%[wx] row_newbcast:%[k] row_mask:0xf bank_mask:0xf\n\t"
#define SC_IP_B(c) "v_fmac_f64_dpp %[" #c "y], %[" #c "y], -%[wx] row_newbcast:%[k] row_mask:0xf bank_mask:0xf\n\t"
#define SC_IP_C(c) "v_fmac_f64_dpp %[" #c "x], %[" #c "y], %[wy] row_newbcast:%[k] row_mask:0xf bank_mask:0xf\n\t"
#define SC_IP_D(c) "v_fmac_f64_dpp %[" #c "y], %[" #c "x], -%[wy] row_newbcast:%[k] row_mask:0xf bank_mask:0xf\n\t"
#define SC_IP_IO(c, v) [c##x] "+v"(v.x), [c##y] "+v"(v.y)
#define SC_IP_IN [wx] "v"(w.x), [wy] "v"(w.y), [k] "n"(K)
template <int K>
__device__ __forceinline__ void cfnma_inplace(cplx &a, const cplx &w) {
    asm volatile(SC_IP_A(a) SC_IP_B(a) "s_nop 1\n\t" SC_IP_C(a) "s_nop 1\n\t" SC_IP_D(a) : SC_IP_IO(a, a) : SC_IP_IN);
}
template <int K>
__device__ __forceinline__ void cfnma_inplace(cplx &a, cplx &b, const cplx &w) {
    asm volatile(SC_IP_A(a) SC_IP_A(b) SC_IP_B(a) SC_IP_B(b) "s_nop 0\n\t" SC_IP_C(a) SC_IP_C(b) "s_nop 0\n\t" SC_IP_D(a) SC_IP_D(b)
                 : SC_IP_IO(a, a), SC_IP_IO(b, b) : SC_IP_IN);
}
template <int K>
__device__ __forceinline__ void cfnma_inplace(cplx &a, cplx &b, cplx &c, const cplx &w) {
    asm volatile(SC_IP_A(a) SC_IP_A(b) SC_IP_A(c) SC_IP_B(a) SC_IP_B(b) SC_IP_B(c)
                 SC_IP_C(a) SC_IP_C(b) SC_IP_C(c) SC_IP_D(a) SC_IP_D(b) SC_IP_D(c)
                 : SC_IP_IO(a, a), SC_IP_IO(b, b), SC_IP_IO(c, c) : SC_IP_IN);
}
template <int K>
__device__ __forceinline__ void cfnma_inplace(cplx &a, cplx &b, cplx &c, cplx &d, const cplx &w) {
    asm volatile(SC_IP_A(a) SC_IP_A(b) SC_IP_A(c) SC_IP_A(d) SC_IP_B(a) SC_IP_B(b) SC_IP_B(c) SC_IP_B(d)
                 SC_IP_C(a) SC_IP_C(b) SC_IP_C(c) SC_IP_C(d) SC_IP_D(a) SC_IP_D(b) SC_IP_D(c) SC_IP_D(d)
                 : SC_IP_IO(a, a), SC_IP_IO(b, b), SC_IP_IO(c, c), SC_IP_IO(d, d) : SC_IP_IN);
}
template <int K>
__device__ __forceinline__ void cfnma_inplace(cplx &a, cplx &b, cplx &c, cplx &d, cplx &e, const cplx &w) {
    asm volatile(SC_IP_A(a) SC_IP_A(b) SC_IP_A(c) SC_IP_A(d) SC_IP_A(e) SC_IP_B(a) SC_IP_B(b) SC_IP_B(c) SC_IP_B(d) SC_IP_B(e)
                 SC_IP_C(a) SC_IP_C(b) SC_IP_C(c) SC_IP_C(d) SC_IP_C(e) SC_IP_D(a) SC_IP_D(b) SC_IP_D(c) SC_IP_D(d) SC_IP_D(e)
                 : SC_IP_IO(a, a), SC_IP_IO(b, b), SC_IP_IO(c, c), SC_IP_IO(d, d), SC_IP_IO(e, e) : SC_IP_IN);
}
// elements I0 .. I1-1 of `v`, in groups of three (the last group takes what is left: 1 .. 5)
template <int K, int I0, int I1, int N>
__device__ __forceinline__ void cfnma_inplace_range(cplx (&v)[N], const cplx &w) {
    constexpr int n = I1 - I0;
    if constexpr (n == 1) cfnma_inplace<K>(v[I0], w);
    else if constexpr (n == 2) cfnma_inplace<K>(v[I0], v[I0 + 1], w);
    else if constexpr (n == 3) cfnma_inplace<K>(v[I0], v[I0 + 1], v[I0 + 2], w);
    else if constexpr (n == 4) cfnma_inplace<K>(v[I0], v[I0 + 1], v[I0 + 2], v[I0 + 3], w);
    else if constexpr (n == 5) cfnma_inplace<K>(v[I0], v[I0 + 1], v[I0 + 2], v[I0 + 3], v[I0 + 4], w);
    else if constexpr (n > 5) {
        cfnma_inplace<K>(v[I0], v[I0 + 1], v[I0 + 2], w);
        cfnma_inplace_range<K, I0 + 3, I1>(v, w);
    }
}
// ---- rows of a trajectory's monodromy blocks staged in LDS by LDS-DMA (global_load_lds_dwordx4), read back one row per lane ----
typedef double lin_d2v __attribute__((ext_vector_type(2)));

// row r of two blocks (DD doubles apart) of the wavefront's buffer -> registers; the reads are inline assembly so that
// the compiler does not put a vmcnt(0) (it cannot tell these reads from the LDS-DMA requests in flight) in front of them
template <int D, int I>
__device__ __forceinline__ void lin_lds_row_reads(unsigned addr, lin_d2v (&v)[D]) {
    if constexpr (I < D / 2) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(v[I]) : "v"(addr), "n"(16 * I));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(v[D / 2 + I]) : "v"(addr), "n"(8 * D * D + 16 * I));
        lin_lds_row_reads<D, I + 1>(addr, v);
    }
}
template <int D>
__device__ __forceinline__ void lin_lds_rows(unsigned addr, double (&Tq)[D], double (&Tp)[D]) {
    lin_d2v v[D];
    lin_lds_row_reads<D, 0>(addr, v);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < D; ++i) asm volatile("" : "+v"(v[i]));        // consumers are ordered behind the wait
#pragma unroll
    for (int i = 0; i < D / 2; ++i) {
        Tq[2 * i] = v[i].x; Tq[2 * i + 1] = v[i].y;
        Tp[2 * i] = v[D / 2 + i].x; Tp[2 * i + 1] = v[D / 2 + i].y;
    }
}


// value of lane K of the row (32-bit)
template <int K>
__device__ __forceinline__ int bc_i32(int v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xF, 0xF, true);
}
// 1/z with a hardware reciprocal refined by two Newton steps (full fp64 accuracy for normal |z|^2)
__device__ __forceinline__ cplx c_inv_newton(cplx z) {
    const double x = c_abs2(z);
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return c_make(z.x * r, -z.y * r);
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

// |pivot|^2 more than 2^8 below the largest candidate partial pivoting could have taken: the fixed pivot order is not
// good enough for this matrix (the caller hands the trajectory to a fully pivoted kernel)
__device__ __forceinline__ bool weak_pivot_keys(int key_pivot, int key_max) { return key_max - key_pivot > (8 << 20); }

// det of the N x N complex matrix held one ROW per lane (lane i: m[0..N); rows >= N must be zero), by Gaussian elimination
// in the FIXED pivot order 0 .. N-1: the pivot row is a static lane, every update is a fused broadcast multiply-add
// (cfnma_inplace), nothing goes through LDS.  `weak` is set (in every lane of the row) if a pivot is zero or more than a
// factor 16 below the largest entry of its column among the rows still to be eliminated.  m is destroyed.
template <int N>
__device__ __forceinline__ cplx det_rows_fixed_order(cplx (&m)[N], int r, int &weak) {
    cplx det = c_make(1.0, 0.0);
    sfor<0, N>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const int key = (r >= k && r < N) ? __double2hiint(c_abs2(m[k])) : 0;
        const int key_max = row_max(key), key_piv = bc_i32<k>(key);
        if (weak_pivot_keys(key_piv, key_max) || key_piv == 0) weak = 1;
        const cplx piv = c_make(bc<k>(m[k].x), bc<k>(m[k].y));
        det = c_mul(det, piv);
        if constexpr (k + 1 < N) {
            const cplx f = c_mul(m[k], c_inv_newton(piv));
            const cplx mult = c_make(r > k ? f.x : 0.0, r > k ? f.y : 0.0);      // rows <= k (the pivot lane among them) stay
            dpp_guard(m);
            cfnma_inplace_range<k, k + 1, N>(m, mult);
        }
    });
    return det;
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
