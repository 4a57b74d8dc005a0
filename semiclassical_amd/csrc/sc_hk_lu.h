// Register-resident determinant of the HK prefactor matrix by block-pivoted Gaussian elimination: the pieces shared by
// the separable fast kernels (sc_hk_step_sd.hip: one 256-thread workgroup per trajectory; sc_hk_step_ws.hip: wave-
// specialised workgroups in which 256-thread groups eliminate while another group streams).  See sc_hk_step_sd.hip for
// the description of the scheme.
#pragma once
#include "sc_common.h"

// -DLU_PIVOT_CLOCK (variant library, tools/lu_pivot_clock.py): the winner lane of every pivot step of workgroup 0 writes the shader
// clock to g_lu_clock[item][16 KB + kt][2] for the first 64 items; [0]: the owner's poll of the previous step returned, [1]: its
// first-slot update is done (publish_pivot_row starts)
#ifdef LU_PIVOT_CLOCK
static __device__ unsigned long long *g_lu_clock = nullptr;
#endif

namespace {

template <int CTRL>
__device__ __forceinline__ int dpp_mov_i32(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}

__device__ __forceinline__ int row16_max_i32(int v) {
    asm("s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf"
        : "+v"(v));
    return v;
}

// value of `v` in lane `src` (wave-uniform index) as a wave-uniform scalar
__device__ __forceinline__ double readlane_f64(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// 1/z with a hardware reciprocal refined by two Newton steps (full fp64 accuracy for normal |z|^2)
__device__ __forceinline__ cplx c_inv_fast(cplx z) {
    const double x = c_abs2(z);
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return c_make(z.x * r, -z.y * r);
}

struct PivotRecord {        // published by the owner of row k together with the scaled row
    int col, pad;           // (unused); tag = (seq << 4) | pivot lane: ONE word tells a consumer that the row is there and which column
};                          // of the diagonal block it eliminates (one LDS read and two VALU instructions less per pivot and wave)

// Pivot-column entries of the N = NR-KB live row slots from lane pl (wave-uniform, run-time) of every 16-lane DPP
// row: DPP row_newbcast (no LDS traffic; 64-bit DPP moves exist on gfx90a+ exactly for this control).  The lane is
// an immediate of the instruction, so there are 16 leaves of 2N moves each and a computed jump (s_setpc_b64) to
// the leaf of pl; a leaf is 2N*8 + 4 bytes (s_branch to the end).
#define SC_DPP_MOV(o, i, P) "v_mov_b64_dpp %[" #o "], %[" #i "] row_newbcast:" #P " row_mask:0xf bank_mask:0xf\n\t"
#define SC_LEAF1(P) SC_DPP_MOV(ox0, ix0, P) SC_DPP_MOV(oy0, iy0, P) "s_branch .Lend_%=\n\t"
#define SC_LEAF2(P) SC_DPP_MOV(ox0, ix0, P) SC_DPP_MOV(oy0, iy0, P) SC_DPP_MOV(ox1, ix1, P) SC_DPP_MOV(oy1, iy1, P) \
    "s_branch .Lend_%=\n\t"
#define SC_LEAF3(P) SC_DPP_MOV(ox0, ix0, P) SC_DPP_MOV(oy0, iy0, P) SC_DPP_MOV(ox1, ix1, P) SC_DPP_MOV(oy1, iy1, P) \
    SC_DPP_MOV(ox2, ix2, P) SC_DPP_MOV(oy2, iy2, P) "s_branch .Lend_%=\n\t"
#define SC_LEAF4(P) SC_DPP_MOV(ox0, ix0, P) SC_DPP_MOV(oy0, iy0, P) SC_DPP_MOV(ox1, ix1, P) SC_DPP_MOV(oy1, iy1, P) \
    SC_DPP_MOV(ox2, ix2, P) SC_DPP_MOV(oy2, iy2, P) SC_DPP_MOV(ox3, ix3, P) SC_DPP_MOV(oy3, iy3, P) "s_branch .Lend_%=\n\t"
#define SC_LEAF5(P) SC_DPP_MOV(ox0, ix0, P) SC_DPP_MOV(oy0, iy0, P) SC_DPP_MOV(ox1, ix1, P) SC_DPP_MOV(oy1, iy1, P) \
    SC_DPP_MOV(ox2, ix2, P) SC_DPP_MOV(oy2, iy2, P) SC_DPP_MOV(ox3, ix3, P) SC_DPP_MOV(oy3, iy3, P)              \
    SC_DPP_MOV(ox4, ix4, P) SC_DPP_MOV(oy4, iy4, P) "s_branch .Lend_%=\n\t"
#define SC_LEAF6(P) SC_DPP_MOV(ox0, ix0, P) SC_DPP_MOV(oy0, iy0, P) SC_DPP_MOV(ox1, ix1, P) SC_DPP_MOV(oy1, iy1, P) \
    SC_DPP_MOV(ox2, ix2, P) SC_DPP_MOV(oy2, iy2, P) SC_DPP_MOV(ox3, ix3, P) SC_DPP_MOV(oy3, iy3, P)              \
    SC_DPP_MOV(ox4, ix4, P) SC_DPP_MOV(oy4, iy4, P) SC_DPP_MOV(ox5, ix5, P) SC_DPP_MOV(oy5, iy5, P) "s_branch .Lend_%=\n\t"
#define SC_LEAVES(L) L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15)
#define SC_JUMP(BYTES)                                                                           \
    "s_getpc_b64 vcc\n"                                                                          \
    ".Lbase_%=:\n\t"                                                                             \
    "s_mul_i32 %[t], %[pl], " #BYTES "\n\t"                                                      \
    "s_add_u32 vcc_lo, vcc_lo, %[t]\n\t"                                                         \
    "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"                                                           \
    "s_add_u32 vcc_lo, vcc_lo, .Lleaf0_%=-.Lbase_%=\n\t"                                         \
    "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"                                                           \
    "s_setpc_b64 vcc\n"                                                                          \
    ".Lleaf0_%=:\n\t"
#define SC_IN(n) [ix##n] "v"(m##n.x), [iy##n] "v"(m##n.y)
#define SC_OUT(n) [ox##n] "=&v"(c##n.x), [oy##n] "=&v"(c##n.y)

__device__ __forceinline__ void column_fetch_n(int pl, cplx m0, cplx &c0) {
    int t;
    asm volatile(SC_JUMP(20) SC_LEAVES(SC_LEAF1) ".Lend_%=:\n" : SC_OUT(0), [t] "=&s"(t) : SC_IN(0), [pl] "s"(pl) : "vcc", "scc");
}
__device__ __forceinline__ void column_fetch_n(int pl, cplx m0, cplx m1, cplx &c0, cplx &c1) {
    int t;
    asm volatile(SC_JUMP(36) SC_LEAVES(SC_LEAF2) ".Lend_%=:\n"
                 : SC_OUT(0), SC_OUT(1), [t] "=&s"(t) : SC_IN(0), SC_IN(1), [pl] "s"(pl) : "vcc", "scc");
}
__device__ __forceinline__ void column_fetch_n(int pl, cplx m0, cplx m1, cplx m2, cplx &c0, cplx &c1, cplx &c2) {
    int t;
    asm volatile(SC_JUMP(52) SC_LEAVES(SC_LEAF3) ".Lend_%=:\n"
                 : SC_OUT(0), SC_OUT(1), SC_OUT(2), [t] "=&s"(t) : SC_IN(0), SC_IN(1), SC_IN(2), [pl] "s"(pl) : "vcc", "scc");
}
__device__ __forceinline__ void column_fetch_n(int pl, cplx m0, cplx m1, cplx m2, cplx m3, cplx &c0, cplx &c1, cplx &c2,
                                               cplx &c3) {
    int t;
    asm volatile(SC_JUMP(68) SC_LEAVES(SC_LEAF4) ".Lend_%=:\n"
                 : SC_OUT(0), SC_OUT(1), SC_OUT(2), SC_OUT(3), [t] "=&s"(t)
                 : SC_IN(0), SC_IN(1), SC_IN(2), SC_IN(3), [pl] "s"(pl) : "vcc", "scc");
}

__device__ __forceinline__ void column_fetch_n(int pl, cplx m0, cplx m1, cplx m2, cplx m3, cplx m4, cplx &c0, cplx &c1, cplx &c2,
                                               cplx &c3, cplx &c4) {
    int t;
    asm volatile(SC_JUMP(84) SC_LEAVES(SC_LEAF5) ".Lend_%=:\n"
                 : SC_OUT(0), SC_OUT(1), SC_OUT(2), SC_OUT(3), SC_OUT(4), [t] "=&s"(t)
                 : SC_IN(0), SC_IN(1), SC_IN(2), SC_IN(3), SC_IN(4), [pl] "s"(pl) : "vcc", "scc");
}
__device__ __forceinline__ void column_fetch_n(int pl, cplx m0, cplx m1, cplx m2, cplx m3, cplx m4, cplx m5, cplx &c0, cplx &c1,
                                               cplx &c2, cplx &c3, cplx &c4, cplx &c5) {
    int t;
    asm volatile(SC_JUMP(100) SC_LEAVES(SC_LEAF6) ".Lend_%=:\n"
                 : SC_OUT(0), SC_OUT(1), SC_OUT(2), SC_OUT(3), SC_OUT(4), SC_OUT(5), [t] "=&s"(t)
                 : SC_IN(0), SC_IN(1), SC_IN(2), SC_IN(3), SC_IN(4), SC_IN(5), [pl] "s"(pl) : "vcc", "scc");
}

template <int NR, int KB>
__device__ __forceinline__ void column_fetch(const cplx (&m)[NR][NR], cplx (&c)[NR], int pl) {
    pl = __builtin_amdgcn_readfirstlane(pl);
    if constexpr (NR - KB == 1) column_fetch_n(pl, m[KB][KB], c[KB]);
    if constexpr (NR - KB == 2) column_fetch_n(pl, m[KB][KB], m[KB + 1][KB], c[KB], c[KB + 1]);
    if constexpr (NR - KB == 3) column_fetch_n(pl, m[KB][KB], m[KB + 1][KB], m[KB + 2][KB], c[KB], c[KB + 1], c[KB + 2]);
    if constexpr (NR - KB == 4)
        column_fetch_n(pl, m[KB][KB], m[KB + 1][KB], m[KB + 2][KB], m[KB + 3][KB], c[KB], c[KB + 1], c[KB + 2], c[KB + 3]);
    if constexpr (NR - KB == 5)
        column_fetch_n(pl, m[KB][KB], m[KB + 1][KB], m[KB + 2][KB], m[KB + 3][KB], m[KB + 4][KB], c[KB], c[KB + 1], c[KB + 2],
                       c[KB + 3], c[KB + 4]);
    if constexpr (NR - KB == 6)
        column_fetch_n(pl, m[KB][KB], m[KB + 1][KB], m[KB + 2][KB], m[KB + 3][KB], m[KB + 4][KB], m[KB + 5][KB], c[KB], c[KB + 1],
                       c[KB + 2], c[KB + 3], c[KB + 4], c[KB + 5]);
}

// The 16 lanes that own row k = 16*KB + kt pick the pivot column among the live columns of the diagonal block,
// scale the row by 1/pivot and publish it: row -> rowbuf[kt], pivot -> pivrec[kt], and LAST the record's tag
// (= seq), which the consumers poll.  LDS operations of one wave execute in issue order, so a consumer that sees
// the tag sees the row.  Runs inside `if (ti == kt)`.  The winner lane stores the (signed) pivot to detbuf[16 KB + kt] in
// LDS (a plain store: round 4; it used to multiply it into a per-row-group product, a read-modify-write on the owner's
// path) -- lu_partial_products / finish_determinant multiply them at the end -- and a zero pivot sets bit 1 of *weak: no
// thread carries the determinant or a singularity flag in registers.  detbuf[0..16 NR) must hold 1 before the first block.
template <int NR, int KB, int RW>
__device__ __forceinline__ void publish_pivot_row(const cplx (&m)[NR][NR], cplx *detbuf, bool live, int kt, int seq,
                                                  cplx (*rowbuf)[RW], PivotRecord *pivrec, int *weak, int tid) {
    const int tj = tid & 15, lane = tid & 63;
    // key = upper 26 bits of |a_kj|^2 (as an integer) | (15 - tj)
    const int blk = (__double2hiint(c_abs2(m[KB][KB])) & ~15) | (15 - tj);
    int key_blk = live ? blk : -1;
    // every lane inverts its own in-block candidate while the search runs; the winner's is used
    const cplx myinv = c_inv_fast(m[KB][KB]);
    key_blk = row16_max_i32(key_blk);
    const int pl = 15 - (key_blk & 15);
    const int src = __builtin_amdgcn_readfirstlane((lane & ~15) | pl);
    const cplx inv = c_make(readlane_f64(myinv.x, src), readlane_f64(myinv.y, src));
    // no masking of the columns that are not live: their entries in this row are exactly zero already (a pivot column is
    // cleared in every remaining row by its own step: a - a * 1), and the pivot column itself gets r = 1, which clears it
    rowbuf[kt][16 * KB + tj] = c_mul(m[KB][KB], inv);
#pragma unroll
    for (int rb = KB + 1; rb < NR; ++rb) rowbuf[kt][16 * rb + tj] = c_mul(m[KB][rb], inv);
    __asm__ volatile("" ::: "memory");
    // the tag FIRST: everything below is behind the hand-over (the per-pivot clocks of tools/lu_pivot_clock.py show the workgroup
    // waiting on this chain: the zero-pivot test alone, moved behind the tag, was worth 2.8 % of the kernel)
    if (tj == pl) {
        __hip_atomic_store(&pivrec[kt].pad, (seq << 4) | pl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef LU_PIVOT_CLOCK
        if (g_lu_clock && blockIdx.x == 0 && ((seq - 1 - KB) >> 2) < 64) g_lu_clock[(((seq - 1 - KB) >> 2) * 64 + 16 * KB + kt) * 4 + 2] = clock64();
#endif
    }
    __asm__ volatile("" ::: "memory");
    // sign of the column choice: the pivot column is the p-th of the live ones (p = live columns to its left; only the
    // 16 owner lanes are active, so the ballot holds exactly their `live` bits) = p adjacent transpositions
    const unsigned long long lm = __ballot(live);
    const int flip = (__popcll(lm & ((1ull << src) - 1ull)) & 1) << 31;
    if (tj == pl) {                                       // the winner stores the signed pivot
        const cplx piv = c_make(__hiloint2double(__double2hiint(m[KB][KB].x) ^ flip, __double2loint(m[KB][KB].x)),
                                __hiloint2double(__double2hiint(m[KB][KB].y) ^ flip, __double2loint(m[KB][KB].y)));
        detbuf[16 * KB + kt] = piv;
        if (m[KB][KB].x == 0.0 && m[KB][KB].y == 0.0) atomicOr(weak, 2);      // singular: det = 0
    }
    __asm__ volatile("" ::: "memory");
    // |pivot|^2 more than 2^8 below some |a_kj|^2 outside the block: the pivoted fallback redoes the trajectory
    if (KB + 1 < NR) {
        int key_out = -1;
#pragma unroll
        for (int rb = KB + 1; rb < NR; ++rb) key_out = max(key_out, __double2hiint(c_abs2(m[KB][rb])));
        if ((key_out & ~15) - (key_blk & ~15) > (8 << 20)) atomicOr(weak, 1);
    }
}

// All elimination steps of the diagonal block KB.  Within block KB only the columns of slot KB are consumed as
// pivots: `live` (per thread) says whether column (slot KB, lane tj) is still available; slots rb > KB are untouched,
// slots rb < KB are finished.  Padded columns (j >= D) hold zeros and can never win the magnitude search unless the
// whole row is zero (= singular: flagged, the arithmetic runs on with inf/nan and the result is discarded).
// No barrier inside the block: the four waves run the 16 steps as a dataflow pipeline.  A wave waits for row kt by
// polling the tag of pivrec[kt] (reading the record and the row in the same batch), fetches the pivot-column entries
// with DPP, updates row slot KB first so that the 16 lanes owning row kt+1 can search and publish at once, and only
// then does the rest of its rank-1 update.  The chain owner(kt) -> owner(kt+1) is the critical path; the bulk of
// the update floats beside it.  `barrier()` in front of the block is the caller's choice: a wave owns every fourth
// pivot step, so when step s is published every wave has consumed step s - 4, and a ring entry is rewritten 16 steps after
// its last use -- only a caller whose waves may lag by a whole block (no pivot ownership in a partial block) needs it.
// The pivot ORDER inside a block is kt = 0..15 with thread index ti = 4 j + w (j = 16-lane row of the wave, w = wave):
// consecutive pivots are owned by different waves.  The matrix ROW a thread holds is trow = 4 w + j (a wave streams
// four consecutive rows, see the kernel), i.e. pivot step kt eliminates row 4 (kt & 3) + (kt >> 2) of the block; steps
// whose row lies beyond D are skipped.  The order in which rows are eliminated does not change the determinant.
__device__ __forceinline__ bool pivot_step_valid(int kt, int nk) { return 4 * (kt & 3) + (kt >> 2) < nk; }

// `tid` = index of the thread inside its 256-thread elimination group (= threadIdx.x when the group is the workgroup),
// `barrier()` synchronises the four wavefronts of the group.
// `detbuf`: the 16 NR signed pivots in LDS (see publish_pivot_row), all 1 before block 0.
template <int NR, int KB, int RW, class Barrier>
__device__ __forceinline__ void eliminate_block(cplx (&m)[NR][NR], cplx *detbuf, int D, int seq,
                                                cplx (*rowbuf)[RW], PivotRecord *pivrec, int *weak, int tid,
                                                Barrier &&barrier) {
    const int ti = ((tid >> 4) & 3) * 4 + (tid >> 6), tj = tid & 15;
    // every caller instantiates NR = ceil(D / 16): only the LAST diagonal block can be partial, the others run their 16 steps
    // without the (scalar, but serial) search for the next valid step
    constexpr bool FULL = KB + 1 < NR;
    const int nk = FULL ? 16 : min(16, D - 16 * KB);
    bool live = FULL ? true : 16 * KB + tj < D;
    barrier();
    if (ti == 0) publish_pivot_row<NR, KB, RW>(m, detbuf, live, 0, seq, rowbuf, pivrec, weak, tid);
    for (int kt = 0; kt < 16; ++kt) {
        if (!FULL && !pivot_step_valid(kt, nk)) continue;
        int next = kt + 1;
        if (!FULL) while (next < 16 && !pivot_step_valid(next, nk)) ++next;
        int tag;
        cplx r[NR];
        for (;;) {
            tag = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&pivrec[kt].pad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            __asm__ volatile("" ::: "memory");
#pragma unroll
            for (int rb = KB; rb < NR; ++rb) r[rb] = rowbuf[kt][16 * rb + tj];
            __asm__ volatile("" ::: "memory");
            if ((tag >> 4) == seq) break;
        }
        const int pl = tag & 15;
#ifdef LU_PIVOT_CLOCK
        if (g_lu_clock && blockIdx.x == 0 && next < 16 && ti == next && tj == 0 && ((seq - 1 - KB) >> 2) < 64)
            g_lu_clock[(((seq - 1 - KB) >> 2) * 64 + 16 * KB + next) * 4 + 0] = clock64();
#endif
        live = live && tj != pl;
        cplx c[NR];
        // rows of this block that have been pivot rows already are updated like the others: nothing reads them again
        column_fetch<NR, KB>(m, c, pl);
#pragma unroll
        for (int rb = KB; rb < NR; ++rb) m[KB][rb] = c_fnma(c[KB], r[rb], m[KB][rb]);
#ifdef LU_PIVOT_CLOCK
        if (g_lu_clock && blockIdx.x == 0 && next < 16 && ti == next && tj == 0 && ((seq - 1 - KB) >> 2) < 64)
            g_lu_clock[(((seq - 1 - KB) >> 2) * 64 + 16 * KB + next) * 4 + 1] = clock64();
#endif
        if (next < 16 && ti == next) publish_pivot_row<NR, KB, RW>(m, detbuf, live, next, seq, rowbuf, pivrec, weak, tid);
#pragma unroll
        for (int ra = KB + 1; ra < NR; ++ra) {
#pragma unroll
            for (int rb = KB; rb < NR; ++rb) m[ra][rb] = c_fnma(c[ra], r[rb], m[ra][rb]);
        }
    }
}

// The determinant from the signed pivots in detbuf[16 * KB + kt] (slots of skipped steps hold 1), after the barrier that ends the
// elimination: threads 0..15 (one wavefront: LDS operations of a wave execute in order) multiply the NR pivots of their step index
// into detbuf[0..16), thread 0 multiplies those 16 and applies the sign of the ROW order.  The signs of the column choices are in
// the pivots already.  Call lu_partial_products from the first 16 threads, then finish_determinant from thread 0.
template <int NR>
__device__ __forceinline__ void lu_partial_products(cplx *detbuf, int t) {
    cplx p = detbuf[t];
#pragma unroll
    for (int kb = 1; kb < NR; ++kb) p = c_mul(p, detbuf[16 * kb + t]);
    detbuf[t] = p;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
}
__device__ __forceinline__ cplx finish_determinant(const cplx *detbuf, bool rows_odd) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    cplx det = detbuf[0];
#pragma unroll
    for (int g = 1; g < 16; ++g) det = c_mul(det, detbuf[g]);
    return rows_odd ? c_make(-det.x, -det.y) : det;
}
// Sign of the order in which the rows are eliminated (pivot step kt of a block takes row 4 (kt & 3) + (kt >> 2), steps
// beyond the last row are skipped): true if the sequence is an odd permutation of the natural order.  Depends on D
// only -- evaluated once per kernel.
__device__ __forceinline__ bool row_order_is_odd(int D) {
    int inversions = 0;
    for (int kb = 0; 16 * kb < D; ++kb) {
        const int nk = min(16, D - 16 * kb);
        for (int k1 = 0; k1 < 16; ++k1)
            for (int k2 = k1 + 1; k2 < 16; ++k2) {
                const int r1 = 4 * (k1 & 3) + (k1 >> 2), r2 = 4 * (k2 & 3) + (k2 >> 2);
                if (r1 < nk && r2 < nk && r1 > r2) ++inversions;
            }
    }
    return (inversions & 1) != 0;
}

}  // namespace
