// Fast path of the fused Herman-Kluk step for SEPARABLE potentials (diagonal Hessian) with DIAGONAL width
// matrices and D <= 64 -- the anharmonic adiabatic-shift configurations of BASELINE.json.
//
// One 256-thread workgroup per trajectory; 1024 persistent workgroups draw trajectories from a device-side cursor.
// Threads form a 16 x 16 grid; thread (trow, tj) owns the elements (a, b) = (16*ra + trow, 16*rb + tj), ra, rb < NR =
// ceil(D/16), of all four monodromy blocks and of the complex prefactor matrix.  Wave w holds the rows trow = 4w .. 4w+3
// of every 16-row slot (contiguous memory in the tiled storage order: a wave instruction covers 512 bytes), while the
// pivot ORDER inside a slot interleaves the waves (consecutive pivots sit in DIFFERENT waves, which is what lets the
// elimination run as a pipeline).  The whole 4*D*D*8-byte state of the trajectory is read once and written once.
//
//   modes    (hk_modes_kernel, one wavefront per trajectory, lane = mode) RK4 of (q_a, p_a); S and <T+V> by wave
//            reductions.  With a diagonal Hessian the monodromy elements (Mqq,Mpq)_ab and (Mqp,Mpp)_ab obey, for
//            every b, the same linear 2x2 system with the stage Hessians h_s[a]; its RK4 step is the 2x2 matrix
//            P_a, obtained by pushing the unit vectors through the reference's stage formula.  P_a -> st.work.
//                                                               (propagators.py:86-119, 313-383; potentials.py)
//   phase B  every (a,b): (Mqq,Mpq)' = P_a (Mqq,Mpq), (Mqp,Mpp)' = P_a (Mqp,Mpp), store back, and form
//            mat_ab = 1/2[ st_a/si_b Mqq + si_b/st_a Mpp - i hbar st_a si_b Mqp + i/hbar Mpq/(st_a si_b) ]
//            in registers.  Raw buffer instructions: scalar tile / plane offset + ONE per-thread 32-bit offset;
//            threads outside the matrix are cut off by the resource's range check (no branches).
//                                                                            (propagators.py:969-986)
//   phase C  c2 = det(mat) by Gaussian elimination with the matrix held in REGISTERS (NR*NR complex per
//            thread).  Rows are eliminated in a fixed order; the pivot COLUMN of a row is chosen by magnitude
//            among the live columns of the diagonal 16-column block (threshold pivoting: the 16 consecutive
//            lanes that own the row search it with integer keys and DPP rotations).  Because pivots stay inside
//            the diagonal block, finished row AND column blocks drop out statically.  The owner scales the row
//            by 1/pivot and publishes it through LDS; the pivot-column entries a thread needs sit in its own
//            16-lane group and are fetched with DPP row_newbcast.  The four waves are NOT barrier-coupled:
//            consumers poll a tag in the pivot record (see eliminate_block); one barrier in front of the first
//            block and one behind the last.  Pivot products (with the signs of the column choices) accumulate in
//            LDS per row group; thread 0 multiplies them while the others stream the next trajectory.
//            If the best in-block pivot is more than 16x smaller than the largest live entry of the row, the
//            trajectory is flagged (sc_state.flags) and its determinant is recomputed by the fully pivoted
//            LDS elimination of sc_hk_step.hip in the same stream (never observed for HK matrices so far; the
//            path is exercised by tests/test_hk_gpu.py::test_weak_pivot_fallback).
//            Then the sqrt branch tracker.                   (torch.det, propagators.py:999, 1006-1052)
#include <algorithm>
#include "sc_common.h"
#include "sc_hk_lu.h"
#include "sc_row16.h"

#ifndef SC_SD_DEPTH
#define SC_SD_DEPTH 1      // row slots whose loads are in flight together (1: the schedule up to round 2)
#endif
#ifndef SC_SD_XPREFETCH
#define SC_SD_XPREFETCH 1  // request row slot 0 of the next trajectory before the last diagonal block of the elimination
#endif
#ifndef SC_SD_XPREFETCH2
#define SC_SD_XPREFETCH2 1 // also row slot 1 of the next trajectory (both raw buffers are free during the last diagonal block): -3 %
#endif
#ifndef SC_SD_XPREFETCH_AHEAD
#define SC_SD_XPREFETCH_AHEAD 1   // diagonal blocks of the elimination that run after the next trajectory's first requests
#endif
// cache-policy bits of the block loads / stores (gfx942+: 1 = sc0, 2 = nt, 16 = sc1).  The blocks are touched once per step
// and the state (11.5 GB) is far beyond every cache: non-temporal on BOTH sides is worth 2.6 % (4.69 -> 4.56 ms; loads
// alone: nothing, stores alone: -0.5 %, scope bits: nothing).  TILED layout only: there a wave instruction covers 512
// contiguous bytes; on the 128-byte row segments of the row-major order (unaligned at D = 60) the same hint costs 45 %
// (4.9 -> 7.2 ms), as it does on the 8-byte-per-lane row accesses of the small-D kernels
#ifndef SC_SD_LOAD_AUX
#define SC_SD_LOAD_AUX 2
#endif
#ifndef SC_SD_MULTI_LOAD_AUX            // first sub-step of a multi-step visit
#define SC_SD_MULTI_LOAD_AUX 2
#endif
#ifndef SC_SD_MULTI_MID_AUX             // stores of an intermediate sub-step, loads of the following one
#define SC_SD_MULTI_MID_AUX 0
#endif
#ifndef SC_SD_STORE_AUX
#define SC_SD_STORE_AUX 2
#endif
#ifndef SC_SD_DIRECT_P
#define SC_SD_DIRECT_P 0     // 1: every thread loads the P_a of its rows itself (no LDS hand-over, one barrier less per
                           // trajectory) -- measured 4.65 ms against 4.56 ms: rejected
#endif
#ifndef SC_SD_ABLATE_LU
#define SC_SD_ABLATE_LU 0  // 1: variant library without the elimination (streaming phase alone)
#endif
#ifndef SC_SD_FORCE_FIXUP
#define SC_SD_FORCE_FIXUP 0
#endif
#ifndef SC_SD_BLOCK_BARRIER
#define SC_SD_BLOCK_BARRIER no_barrier     // wg_barrier: a workgroup barrier in front of every diagonal block
#endif

namespace {

typedef unsigned int sc_v2u __attribute__((ext_vector_type(2)));
typedef unsigned int sc_v4u __attribute__((ext_vector_type(4)));

// TILED: the monodromy blocks of a trajectory are stored as 16 x 16 tiles (sc_state.mono_layout = 1, see the header):
// tile (ra, rb) holds its part of Mqq, Mqp, Mpq, Mpp one after the other, each row-major inside the tile.  With the
// row mapping trow = 4 w + j a wave instruction then covers 512 contiguous bytes, the four waves one tile plane, and
// the workgroup walks the 4 D^2 doubles of the trajectory linearly -- measured 14 % more streaming bandwidth than the
// 128-byte row segments of the row-major layout (tools/micro/stream_patterns.hip).
//
// KS > 1 (round 4, sc_hk_step_multi): KS consecutive time steps per VISIT of a trajectory.  The blocks are stored after every
// sub-step as before, but the next sub-step of the same trajectory follows at once, by the same threads at the same addresses:
// its loads hit the L2 / the 256 MB memory-side cache (1024 workgroups hold 118 MB between two sub-steps) instead of HBM
// (tools/micro/revisit.hip, profiles/r4_revisit.txt: a second read-modify-write visit costs 1.1 ms instead of 4.6).  The
// intermediate stores and loads are plain (cacheable), the first load and the last store of a visit non-temporal (the first
// load plain, the last store plain: same time).  Measured (profiles/r4_sd_phases.txt): WITHOUT the elimination the pair kernel
// streams at 2.94 ms per step (one step per visit: 4.39); WITH it 4.40 against 4.63 -- in this mode the kernel is bound by the
// elimination (26-36 us of a workgroup's 39-50 us per item; FP64 VALU issue, 64-71 % busy in the SQ counters), no longer by HBM.
// Load depth 2 / 1.5, earlier prefetch, wave priority for the owner of the next pivot, an early request of thread 0's previous
// determinant: each measured, none faster (spills at 128 VGPRs, or no effect).  Row
// propagators of sub-step ks: M.work[ks][n][4][D] (hk_modes_multi_kernel); determinant and branch sign after sub-step
// ks < KS - 1: M.c2_mid / M.sgn_mid [ks][n], after the last one: the state's own arrays.
struct MultiArgs {
    const double *work;        // [KS][n][4][D]; KS = 1: unused (st.work)
    double *c2_mid, *sgn_mid;  // [KS - 1][n]
    int *unrepaired;           // counts determinants of INTERMEDIATE sub-steps whose in-block pivots were weak: the blocks have
                               // moved on by the time a fix-up launch could recompute them (sc_hk_step_multi's contract)
};

// -DSD_PHASE_CLOCK (variant library, tools/sd_phases.py): lane 0 of every wave of workgroups 0 and 512 writes the 100 MHz wall clock
// at six points of every item (trajectory sub-step) to g_sd_clock[2][4][256][6] -- raw time stamps, no accumulators in registers
#ifdef SD_PHASE_CLOCK
__device__ unsigned long long *g_sd_clock = nullptr;
#define SD_TICK(i)                                                                                                       \
    do {                                                                                                                 \
        if (pc_blk >= 0 && pc_item < 256 && (threadIdx.x & 63) == 0)                                                     \
            g_sd_clock[((pc_blk * 4 + (threadIdx.x >> 6)) * 256 + pc_item) * 6 + (i)] = wall_clock64();                  \
    } while (0)
#else
#define SD_TICK(i) do { } while (0)
#endif

template <int NR, int MINW, bool STEP, bool TILED, int KS = 1>
__global__ __launch_bounds__(256, MINW) void hk_step_sd_kernel(StepArgs A, MultiArgs MA) {
#ifdef SD_PHASE_CLOCK
    const int pc_blk = g_sd_clock == nullptr ? -1 : (blockIdx.x == 0 ? 0 : (blockIdx.x == 512 ? 1 : -1));
    int pc_item = 0;
#endif
    __shared__ double prop[4 * 64];          // P_a = (p11, p12, p21, p22) of row a
    __shared__ double scl[4 * 64];           // st, 1/st, si, 1/si
    __shared__ cplx rowbuf[16][64];
    __shared__ PivotRecord pivrec[16];
    // per-trajectory results of the elimination, double-buffered by trajectory parity: thread 0 finishes trajectory t
    // (product of the partial determinants, branch tracker) while the other waves already stream trajectory t+1
    __shared__ cplx detbuf[2][16 * NR];      // signed pivots, slot 16 KB + kt
    __shared__ int weakbuf[2];               // bit 0: weak in-block pivot (-> pivoted fallback), bit 1: zero pivot
    __shared__ int nextbuf[2];               // what thread 0 drew from the trajectory cursor

    const int D = A.st.dim, DD = D * D, tid = threadIdx.x;
    constexpr bool do_step = STEP;                   // false: prefactor and tracker initialisation only (t = 0)
    constexpr int NCL_BASE = 16 * (NR - 1);          // first column of the last column tile
    if (tid < 64) {
        const bool in = tid < D;
        const double st = in ? A.hk.st[tid] : 1.0, si = in ? A.hk.si[tid] : 1.0;
        if (tid < 16) pivrec[tid].pad = 0;
        if (tid < 2) { nextbuf[tid] = 0; weakbuf[tid] = 0; }      // never read before they are written (RESET BARRIER below); defined anyway
        scl[tid] = st; scl[64 + tid] = 1.0 / st; scl[128 + tid] = si; scl[192 + tid] = 1.0 / si;
        prop[tid] = 1.0; prop[64 + tid] = 0.0; prop[128 + tid] = 0.0; prop[192 + tid] = 1.0;
    }
    __syncthreads();

    const bool rows_odd = row_order_is_odd(D);
    // the thread index is rebuilt per trajectory from the wave number (a scalar) and the lane number (v_mbcnt): no
    // thread-index register lives across the elimination
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    int seq0 = 0;                                  // tags of this workgroup's pivot records: unique per (trajectory, block)
    int par = 0;
    // raw[s][p][rb]: the four planes (Mqq, Mqp, Mpq, Mpp) of row slot ra = s (mod 2); prv: this thread's element of the
    // row propagators P_a (st.work, computed by hk_modes_kernel, "phase A").  With SC_SD_XPREFETCH slot 0 and prv of the
    // NEXT trajectory are requested before the last diagonal block of the current elimination starts (most of the matrix
    // registers are dead by then): the first load round trip of a trajectory hides behind that block.
    double raw[2][4][NR];
    double pv[2][4] = {{1.0, 0.0, 0.0, 1.0}, {1.0, 0.0, 0.0, 1.0}};     // SC_SD_DIRECT_P: P_a of the thread's row of slot ra = s (mod 2)
    double prv = 0.0;
    bool first = true;
    // Trajectories are handed out through a device-side cursor (sc_state.flags[n + 1], zeroed by sc_hk_step): the first
    // gridDim.x statically, every further one to whichever workgroup is ready next.  The trajectories in flight are then
    // always ~gridDim.x CONSECUTIVE ones (static strides let the workgroups drift apart until the accesses are spread over
    // the whole state: measured 5.3 ms against 4.8 ms for compact windows), and nobody waits for a slow workgroup at the
    // end.  Thread 0 draws the next index at the top of a trajectory; it reaches the others through LDS behind the first
    // barrier of the elimination.  Without sc_state.flags: static stride.
    int *cursor = A.st.flags ? A.st.flags + A.st.n + 1 : nullptr;
    int64_t trn = 0;
    for (int64_t tr = blockIdx.x; tr < A.st.n; tr = trn) {
      sfor<0, KS>([&](auto ksc) {
        constexpr int ks = decltype(ksc)::value;                 // sub-step of this visit
        constexpr int aux_load = ks == 0 ? (KS > 1 ? SC_SD_MULTI_LOAD_AUX : SC_SD_LOAD_AUX) : SC_SD_MULTI_MID_AUX;
        constexpr int aux_store = ks == KS - 1 ? SC_SD_STORE_AUX : SC_SD_MULTI_MID_AUX;
        // what is streamed next: the next sub-step of this trajectory, or the first one of the next trajectory
        constexpr int ksn = ks + 1 < KS ? ks + 1 : 0;
        int *weak = &weakbuf[par];
        // Everything derived from the thread index is recomputed per trajectory: hipcc otherwise keeps those values live
        // across the elimination, spills them, and reloads them one by one behind s_waitcnt vmcnt(0) in the load stream
        unsigned ones = ~0u;
        __asm__ volatile("" : "+s"(ones));
        const int tl = (wave_s << 6) | (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
        const int tj = tl & 15, tjl = tj;
        const int til = (tl >> 6) * 4 + ((tl >> 4) & 3);    // wave w holds rows 4w .. 4w+3 of every 16-row slot
        if (tl == 0) *weak = SC_SD_FORCE_FIXUP;             // 1: variant library that hands every trajectory to the fallback
        if (tl < 16 * NR) detbuf[par][tl] = c_make(1.0, 0.0);
        int drawn = 0;
        if (ks == 0 && cursor && tl == 0) drawn = atomicAdd(cursor, 1);
        const int pk = tl >> 6, pa = tl & 63;               // thread -> (row of P, mode) of st.work

        // ---------------- phase B ----------------
        cplx m[NR][NR];
        // All accesses to the trajectory's blocks are raw BUFFER instructions: one resource (base M, 32 D^2 bytes), the
        // tile / plane position as scalar byte offset, the thread's position inside the tile as 32-bit VGPR offset.
        // Threads outside the matrix carry an offset beyond the resource: their loads return 0 and their stores are
        // dropped by the range check, so phase B has no branches and no 64-bit address arithmetic.
        int Dl = D;
        __asm__ volatile("" : "+s"(Dl));                 // tile offsets are recomputed (SALU) per trajectory, not kept in SGPRs
        constexpr unsigned OOB = 0x7fffffffu;
        const bool rowok_last = 16 * (NR - 1) + til < Dl, colok_last = NCL_BASE + tj < Dl;
        // byte offsets of this thread inside a tile: row-major (row, tj) of a D x D plane; tiled: inside a 16-wide / the
        // last (narrower) column tile
        // tiled: 16 bytes per element of a plane pair (Mqq, Mqp) / (Mpq, Mpp)
        const unsigned vo = TILED ? 16u * (unsigned)(til * 16 + tj) : 8u * (unsigned)(til * Dl + tj);
        const unsigned vol = TILED ? 16u * (unsigned)(til * (Dl - NCL_BASE) + tj) : vo;
        auto voffset = [&](int ra, int rb) {
            const unsigned v = rb == NR - 1 ? vol : vo;
            const bool ok = (ra < NR - 1 || rowok_last) && (rb < NR - 1 || colok_last);
            return (int)((ra < NR - 1 && rb < NR - 1) ? v : (ok ? v : OOB));
        };
        auto tile_base = [&](int ra, int rb) {           // scalar byte offset of tile (ra, rb)
            const int nra = min(16, Dl - 16 * ra);
            return 8 * (TILED ? 4 * (16 * ra * Dl + nra * 16 * rb) : 16 * ra * Dl + 16 * rb);
        };
        auto plane_bytes = [&](int ra, int rb) {         // distance between the four blocks of an element
            const int nra = min(16, Dl - 16 * ra), ncb = rb == NR - 1 ? Dl - NCL_BASE : 16;
            return 8 * (TILED ? nra * ncb : Dl * Dl);
        };
        auto resource = [&](int64_t t) {
            return __builtin_amdgcn_make_buffer_rsrc(A.st.mono + t * 4 * (int64_t)Dl * Dl, 0, 32 * Dl * Dl, 0x00020000);
        };
        auto load_slot = [&](auto rac, int64_t t, auto auxc) {
            constexpr int ra = decltype(rac)::value, aux = decltype(auxc)::value;
            const __amdgpu_buffer_rsrc_t rs = resource(t);
#pragma unroll
            for (int rb = 0; rb < NR; ++rb) {
                const int vofs = voffset(ra, rb), base = tile_base(ra, rb), plane = plane_bytes(ra, rb);
                if constexpr (TILED) {                     // one 16-byte load per plane pair
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        const sc_v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, vofs, base + pp * 2 * plane, aux);
                        raw[ra & 1][2 * pp][rb] = __hiloint2double((int)v.y, (int)v.x);
                        raw[ra & 1][2 * pp + 1][rb] = __hiloint2double((int)v.w, (int)v.z);
                    }
                } else {
#pragma unroll
                    for (int pl = 0; pl < 4; ++pl) {
                        const sc_v2u v = __builtin_amdgcn_raw_buffer_load_b64(rs, vofs, base + pl * plane, 0);
                        raw[ra & 1][pl][rb] = __hiloint2double((int)v.y, (int)v.x);
                    }
                }
            }
            if (SC_SD_DIRECT_P && do_step) {
                // the four entries of P_a of this thread's row come straight from st.work (the 16 threads of a row read the
                // same address; rows beyond D: offset beyond the resource, zeros): no LDS hand-over, no barrier
                const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(A.st.work + t * 4 * (int64_t)Dl, 0, 32 * Dl, 0x00020000);
                const unsigned vw = 16 * ra + til < Dl ? 8u * (unsigned)(16 * ra + til) : OOB;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const sc_v2u v = __builtin_amdgcn_raw_buffer_load_b64(rw, (int)vw, 8 * k * Dl, 0);
                    pv[ra & 1][k] = __hiloint2double((int)v.y, (int)v.x);
                }
            }
        };
        auto first_requests = [&](int64_t t, auto ksnc) {    // P_a and row slot 0 of sub-step ksnc of trajectory t
            constexpr int k2 = decltype(ksnc)::value;
            constexpr int aux2 = k2 == 0 ? (KS > 1 ? SC_SD_MULTI_LOAD_AUX : SC_SD_LOAD_AUX) : SC_SD_MULTI_MID_AUX;
            const double *wk = KS > 1 ? MA.work + (int64_t)k2 * A.st.n * 4 * Dl : A.st.work;
            if (!SC_SD_DIRECT_P && do_step && pa < Dl) prv = wk[(t * 4 + pk) * (int64_t)Dl + pa];
            load_slot(std::integral_constant<int, 0>(), t, std::integral_constant<int, aux2>());
            // SC_SD_XPREFETCH2: row slot 1 as well (both raw buffers are free during the last diagonal block)
            if (SC_SD_XPREFETCH2 && NR > 1) load_slot(std::integral_constant<int, (NR > 1 ? 1 : 0)>(), t, std::integral_constant<int, aux2>());
        };
        auto finish_slot = [&](auto rac) {
            constexpr int ra = decltype(rac)::value;
            const __amdgpu_buffer_rsrc_t rs = resource(tr);
            const int al = (16 * ra + til) & 63;
            const double p11 = SC_SD_DIRECT_P ? pv[ra & 1][0] : prop[al], p12 = SC_SD_DIRECT_P ? pv[ra & 1][1] : prop[64 + al];
            const double p21 = SC_SD_DIRECT_P ? pv[ra & 1][2] : prop[128 + al], p22 = SC_SD_DIRECT_P ? pv[ra & 1][3] : prop[192 + al];
            const double sta = scl[al], ista = scl[64 + al];
#pragma unroll
            for (int rb = 0; rb < NR; ++rb) {
                const int vofs = voffset(ra, rb), base = tile_base(ra, rb), plane = plane_bytes(ra, rb);
                double mqq = raw[ra & 1][0][rb], mqp = raw[ra & 1][1][rb], mpq = raw[ra & 1][2][rb], mpp = raw[ra & 1][3][rb];
                if (do_step) {
                    const double nqq = fma(p12, mpq, p11 * mqq), npq = fma(p22, mpq, p21 * mqq);
                    const double nqp = fma(p12, mpp, p11 * mqp), npp = fma(p22, mpp, p21 * mqp);
                    mqq = nqq; mpq = npq; mqp = nqp; mpp = npp;
                    const double out[4] = {mqq, mqp, mpq, mpp};
                    if constexpr (TILED) {
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp) {
                            sc_v4u v;
                            v.x = (unsigned)__double2loint(out[2 * pp]); v.y = (unsigned)__double2hiint(out[2 * pp]);
                            v.z = (unsigned)__double2loint(out[2 * pp + 1]); v.w = (unsigned)__double2hiint(out[2 * pp + 1]);
                            __builtin_amdgcn_raw_buffer_store_b128(v, rs, vofs, base + pp * 2 * plane, aux_store);
                            // STORE-DATA HAZARD (found on MI355X, round 4): a buffer store of more than 8 bytes reads its data
                            // registers over more than one cycle, and a VALU instruction that overwrites them in the next wait
                            // states changes what is stored (lanes 12..15 of every row of the last wave got Im M' instead of
                            // Mpq, in 1 % of the trajectories, only with many workgroups in flight).  hipcc's hazard recogniser
                            // covers the case WITHOUT a scalar offset register only (GCNHazardRecognizer::createsVALUHazard);
                            // these stores have one.  Three wait states, and nothing is scheduled across them.
                            __builtin_amdgcn_sched_barrier(0);
                            __asm__ volatile("s_nop 2");
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {
#pragma unroll
                        for (int pl = 0; pl < 4; ++pl) {
                            sc_v2u v;
                            v.x = (unsigned)__double2loint(out[pl]); v.y = (unsigned)__double2hiint(out[pl]);
                            __builtin_amdgcn_raw_buffer_store_b64(v, rs, vofs, base + pl * plane, 0);
                        }
                    }
                }
                const int bl = (16 * rb + tjl) & 63;
                const double sib = scl[128 + bl], isib = scl[192 + bl];
                // elements outside the matrix were loaded as zeros: their mat entries are zeros
                m[ra][rb] = c_make(0.5 * (sta * isib * mqq + ista * sib * mpp),
                                   0.5 * (-SC_HBAR * sta * sib * mqp + (1.0 / SC_HBAR) * ista * isib * mpq));
                // the element must exist HERE: hipcc otherwise sinks its computation to the first use in the elimination
                // and carries (spills) the four raw values and the scalings instead
                __asm__ volatile("" : "+v"(m[ra][rb].x), "+v"(m[ra][rb].y));
            }
        };
        // sched_barrier: the stages are scheduled one by one (hipcc otherwise interleaves the stages of this branch-free
        // block until the raw values of three slots are live at once, and spills)
        SD_TICK(0);
        if (first || !SC_SD_XPREFETCH || NR == 1) first_requests(tr, ksc);
        first = false;
        __builtin_amdgcn_sched_barrier(0);
        if (NR > 1 && SC_SD_DEPTH > 1) load_slot(std::integral_constant<int, (NR > 1 ? 1 : 0)>(), tr, std::integral_constant<int, aux_load>());
        __builtin_amdgcn_sched_barrier(0);
        if (do_step && !SC_SD_DIRECT_P) {
            // every wave is past the previous trajectory's phase B (the elimination barriers lie in between), so prop
            // may be overwritten without another barrier
            if (pa < D) prop[64 * pk + pa] = prv;
            __syncthreads();
        }
        sfor<0, NR>([&](auto rac) {
            constexpr int ra = decltype(rac)::value;
            // SC_SD_XPREFETCH2: slots 0 and 1 were requested during the previous elimination (requesting slot 2 before slot 1 is
            // finished as well: 4 spilled registers, 0.5 % slower)
            if (SC_SD_DEPTH == 1 && ra > (SC_SD_XPREFETCH2 ? 1 : 0)) load_slot(rac, tr, std::integral_constant<int, aux_load>());
            if (SC_SD_DEPTH > 1 && ra + 1 < NR && ra > 0) load_slot(std::integral_constant<int, (ra + 1 < NR ? ra + 1 : 0)>(), tr, std::integral_constant<int, aux_load>());
            __builtin_amdgcn_sched_barrier(0);
            finish_slot(rac);
            __builtin_amdgcn_sched_barrier(0);
        });

        // ---------------- phase C: determinant in registers ----------------
        // phase ablation (tools/phase_timing.py) is a COMPILE-time switch of a variant library: a run-time flag here costs the
        // kernel 148 B/lane of scratch and 30 % of its speed
        constexpr bool skip_lu = SC_SD_ABLATE_LU != 0;
        SD_TICK(1);
        if (ks == 0 && cursor && tl == 0) nextbuf[par] = drawn;
        // RESET BARRIER -- unconditional, in every variant of this kernel (with or without the elimination).  It orders
        //   (a) thread 0's nextbuf[par] = drawn against the readfirstlane(nextbuf[par]) of every wave below: the value is
        //       the next trajectory index and goes straight into resource(trn), an unordered read addresses memory
        //       outside the state (the GPU memory fault of round 2's tuning build, docs/NOTEBOOK.md section 8.2);
        //   (b) the resets of *weak and detbuf[par] above against the owner lanes' read-modify-writes in the elimination;
        //   (c) the previous trajectory's last uses of the pivot ring (rowbuf / pivrec) against block 0's first records.
        // Later blocks need no barrier: a wave owns every fourth pivot step, so when step s is published every wave has
        // consumed step s - 4, and a ring entry is rewritten 16 steps after its last use.
        __syncthreads();
        SD_TICK(2);
        auto wg_barrier = [] { __syncthreads(); };
        auto no_barrier = [] {};
        (void)no_barrier; (void)wg_barrier;
        sfor<0, NR>([&](auto kbc) {
            constexpr int KB = decltype(kbc)::value;
            if (ks == 0 && KB == (NR > 1 ? 1 : 0)) { // behind the RESET BARRIER (NR = 1: the value is read after the last barrier)
                if (NR > 1) trn = cursor ? (int64_t)gridDim.x + __builtin_amdgcn_readfirstlane(nextbuf[par]) : tr + gridDim.x;
            }
            if (SC_SD_XPREFETCH && NR > 1 && KB == (NR - SC_SD_XPREFETCH_AHEAD > 1 ? NR - SC_SD_XPREFETCH_AHEAD : 1)) {
                const int64_t tnext = ks + 1 < KS ? tr : trn;
                if (tnext < A.st.n) {
                    first_requests(tnext, std::integral_constant<int, ksn>());
                } else {                            // nothing follows: end the live ranges of the old values
                    prv = 0.0;
#pragma unroll
                    for (int pl = 0; pl < 4; ++pl) {
                        pv[0][pl] = 0.0;
#pragma unroll
                        for (int rb = 0; rb < NR; ++rb) {
                            raw[0][pl][rb] = 0.0;
                            if (SC_SD_XPREFETCH2) raw[1][pl][rb] = 0.0;
                        }
                    }
                }
            }
            if (!skip_lu) {
                if (KB == 0) eliminate_block<NR, KB, 64>(m, detbuf[par], D, seq0 + 1 + KB, rowbuf, pivrec, weak, tl, no_barrier);   // the RESET BARRIER is its barrier
                else eliminate_block<NR, KB, 64>(m, detbuf[par], D, seq0 + 1 + KB, rowbuf, pivrec, weak, tl, SC_SD_BLOCK_BARRIER);
            }
        });
        SD_TICK(3);
        __syncthreads();
        SD_TICK(4);
        if (ks == 0 && NR == 1) trn = cursor ? (int64_t)gridDim.x + __builtin_amdgcn_readfirstlane(nextbuf[par]) : tr + gridDim.x;
        // no barrier after this: the buffers of this parity are next written two visits on
        cplx *c2 = (cplx *)A.st.c2;
        // previous value / destination of this sub-step's determinant and branch sign
        const cplx *c2_in = ks == 0 ? c2 + tr : (const cplx *)MA.c2_mid + ((int64_t)(ks - 1) * A.st.n + tr);
        const double *sg_in = ks == 0 ? A.st.sgn + tr : MA.sgn_mid + ((int64_t)(ks - 1) * A.st.n + tr);
        cplx *c2_out = ks == KS - 1 ? c2 + tr : (cplx *)MA.c2_mid + ((int64_t)ks * A.st.n + tr);
        double *sg_out = ks == KS - 1 ? A.st.sgn + tr : MA.sgn_mid + ((int64_t)ks * A.st.n + tr);
        if (tl < 16) lu_partial_products<NR>(detbuf[par], tl);
        if (ks == KS - 1 && tl == 0 && (*weak & 1) && A.st.flags && !skip_lu) {
            A.st.flags[tr] = 1;                  // c2 / sgn are left to the fully pivoted fallback
            atomicAdd(&A.st.flags[A.st.n], 1);   // lets the fix-up launch return at once when nothing was flagged
        } else if (tl == 0) {
            if (KS > 1 && ks < KS - 1 && (*weak & 1) && !skip_lu) atomicAdd(MA.unrepaired, 1);
            const cplx c2new = (*weak & 2) ? c_make(0.0, 0.0) : finish_determinant(detbuf[par], rows_odd);
            if (do_step) {
                const cplx prev = *c2_in;
                const double sg = *sg_in;
                const bool flip = prev.x < 0.0 && c2new.x < 0.0 && prev.y * c2new.y < 0.0;
                if (KS > 1) *sg_out = flip ? -sg : sg;
                else if (flip) *sg_out = -sg;
            } else {
                *sg_out = 1.0;
            }
            *c2_out = c2new;
        }
        seq0 += 4; par ^= 1;
        SD_TICK(5);
#ifdef SD_PHASE_CLOCK
        ++pc_item;
#endif
      });
    }
}

#ifdef SD_PHASE_CLOCK
}  // namespace
extern "C" int sc_sd_phase_clock(unsigned long long *buf) {       // device buffer of 2*4*256*6 time stamps, or NULL to switch off
    return hipMemcpyToSymbol(HIP_SYMBOL(g_sd_clock), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
namespace {
#endif

// "phase A" as its own launch: one wavefront per trajectory, lane = mode.  RK4 of (q_a, p_a) with the reference's
// stage formula, the action and <T+V> at the k4 stage by wave reductions, and the 2x2 RK4 propagator P_a of the
// monodromy rows (unit vectors pushed through the same stage formula) -> st.work[tr][4][D].
__global__ __launch_bounds__(256) void hk_modes_kernel(StepArgs A) {
    const int D = A.st.dim, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double dt = A.dt, hh = 0.5 * dt, h6 = dt / 6.0;
    __shared__ double wsum[4];
    double esum = 0.0;
    for (int64_t tr = (int64_t)blockIdx.x * 4 + wave; tr < A.st.n; tr += (int64_t)gridDim.x * 4) {
        double *qp = A.st.qp + tr * 2 * D;
        double *pr = A.st.work + tr * 4 * (int64_t)D;
        double red5[5] = {0, 0, 0, 0, 0};
        if (lane < D) {
            const double q = qp[lane], p = qp[D + lane], im = A.pot.inv_mass[lane];
            const double c0 = A.pot.par0[lane], c1 = A.pot.par1 ? A.pot.par1[lane] : 0.0;
            double v, g, h1, h2, h3, h4;
            sep_eval(A.pot.kind, c0, c1, q, v, g, h1);
            const double kq1 = p * im, kp1 = -g;
            red5[0] = 0.5 * p * p * im - v;
            const double q2 = q + hh * kq1, p2 = p + hh * kp1;
            sep_eval(A.pot.kind, c0, c1, q2, v, g, h2);
            const double kq2 = p2 * im, kp2 = -g;
            red5[1] = 0.5 * p2 * p2 * im - v;
            const double q3 = q + hh * kq2, p3 = p + hh * kp2;
            sep_eval(A.pot.kind, c0, c1, q3, v, g, h3);
            const double kq3 = p3 * im, kp3 = -g;
            red5[2] = 0.5 * p3 * p3 * im - v;
            const double q4 = q + dt * kq3, p4 = p + dt * kp3;
            sep_eval(A.pot.kind, c0, c1, q4, v, g, h4);
            const double kq4 = p4 * im, kp4 = -g;
            red5[3] = 0.5 * p4 * p4 * im - v;
            red5[4] = 0.5 * p4 * p4 * im + v;
            qp[lane] = q + h6 * (kq1 + 2.0 * kq2 + 2.0 * kq3 + kq4);
            qp[D + lane] = p + h6 * (kp1 + 2.0 * kp2 + 2.0 * kp3 + kp4);
            double u1 = 1.0, v1 = 0.0, u2 = 0.0, v2 = 1.0;
            rk4_pair(u1, v1, im, h1, h2, h3, h4, dt);
            rk4_pair(u2, v2, im, h1, h2, h3, h4, dt);
            pr[lane] = u1; pr[D + lane] = u2; pr[2 * D + lane] = v1; pr[3 * D + lane] = v2;
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) red5[i] = wave_sum(red5[i]);
        if (lane == 0) {
            A.st.act[tr] += h6 * (red5[0] + 2.0 * red5[1] + 2.0 * red5[2] + red5[3]);
            esum += red5[4];
        }
    }
    if (lane == 0) wsum[wave] = esum;
    __syncthreads();
    if (threadIdx.x == 0 && A.epart) A.epart[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}


// hk_modes_kernel for KS consecutive time steps (sc_hk_step_multi): (q_a, p_a) stay in registers between the sub-steps; row
// propagators of sub-step ks -> work[ks][n][4][D]; (q, p, S) after sub-step ks < KS - 1 -> qp_mid / act_mid [ks][n] (what the
// correlation kernel of time step k + ks + 1 reads), after the last one -> the state; <T+V> sums of sub-step ks -> epart[ks][grid].
// The arithmetic of a sub-step is that of hk_modes_kernel.
struct ModesMultiArgs {
    double *work, *qp_mid, *act_mid;
};
template <int KS>
__global__ __launch_bounds__(256) void hk_modes_multi_kernel(StepArgs A, ModesMultiArgs MM) {
    const int D = A.st.dim, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double dt = A.dt, hh = 0.5 * dt, h6 = dt / 6.0;
    __shared__ double wsum[KS][4];
    double esum[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) esum[ks] = 0.0;
    for (int64_t tr = (int64_t)blockIdx.x * 4 + wave; tr < A.st.n; tr += (int64_t)gridDim.x * 4) {
        double *qp = A.st.qp + tr * 2 * D;
        const bool in = lane < D;
        double q = in ? qp[lane] : 0.0, p = in ? qp[D + lane] : 0.0;
        const double im = in ? A.pot.inv_mass[lane] : 0.0, c0 = in ? A.pot.par0[lane] : 0.0, c1 = (in && A.pot.par1) ? A.pot.par1[lane] : 0.0;
        double act = A.st.act[tr];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            double *pr = MM.work + ((int64_t)ks * A.st.n + tr) * 4 * (int64_t)D;
            double red5[5] = {0, 0, 0, 0, 0};
            if (in) {
                double v, g, h1, h2, h3, h4;
                sep_eval(A.pot.kind, c0, c1, q, v, g, h1);
                const double kq1 = p * im, kp1 = -g;
                red5[0] = 0.5 * p * p * im - v;
                const double q2 = q + hh * kq1, p2 = p + hh * kp1;
                sep_eval(A.pot.kind, c0, c1, q2, v, g, h2);
                const double kq2 = p2 * im, kp2 = -g;
                red5[1] = 0.5 * p2 * p2 * im - v;
                const double q3 = q + hh * kq2, p3 = p + hh * kp2;
                sep_eval(A.pot.kind, c0, c1, q3, v, g, h3);
                const double kq3 = p3 * im, kp3 = -g;
                red5[2] = 0.5 * p3 * p3 * im - v;
                const double q4 = q + dt * kq3, p4 = p + dt * kp3;
                sep_eval(A.pot.kind, c0, c1, q4, v, g, h4);
                const double kq4 = p4 * im, kp4 = -g;
                red5[3] = 0.5 * p4 * p4 * im - v;
                red5[4] = 0.5 * p4 * p4 * im + v;
                q = q + h6 * (kq1 + 2.0 * kq2 + 2.0 * kq3 + kq4);
                p = p + h6 * (kp1 + 2.0 * kp2 + 2.0 * kp3 + kp4);
                double u1 = 1.0, v1 = 0.0, u2 = 0.0, v2 = 1.0;
                rk4_pair(u1, v1, im, h1, h2, h3, h4, dt);
                rk4_pair(u2, v2, im, h1, h2, h3, h4, dt);
                pr[lane] = u1; pr[D + lane] = u2; pr[2 * D + lane] = v1; pr[3 * D + lane] = v2;
                double *qo = ks == KS - 1 ? qp : MM.qp_mid + ((int64_t)ks * A.st.n + tr) * 2 * D;
                qo[lane] = q; qo[D + lane] = p;
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) red5[i] = wave_sum(red5[i]);
            act += h6 * (red5[0] + 2.0 * red5[1] + 2.0 * red5[2] + red5[3]);
            if (lane == 0) {
                if (ks == KS - 1) A.st.act[tr] = act; else MM.act_mid[(int64_t)ks * A.st.n + tr] = act;
                esum[ks] += red5[4];
            }
        }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) if (lane == 0) wsum[ks][wave] = esum[ks];
    __syncthreads();
    if (threadIdx.x < KS && A.epart)
        A.epart[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = wsum[threadIdx.x][0] + wsum[threadIdx.x][1] + wsum[threadIdx.x][2] + wsum[threadIdx.x][3];
}

// ---- D <= 16: one WAVEFRONT per trajectory, everything in registers, no barrier and no LDS hand-off in the elimination.
// Lane (rg, tj) = (lane >> 4, lane & 15) holds column tj of the matrix rows rg, rg + 4, rg + 8, rg + 12 (rows and
// columns beyond D are padded with the identity), for the four monodromy blocks and the prefactor matrix alike; a
// load / store instruction covers four 128-byte row segments.  The modes (RK4 of q, p, S and the row propagators P_a)
// are done by the first D lanes of the same wavefront; P_a reaches the lanes that hold row a through a per-wave LDS
// array.  Elimination, fully unrolled over the pivot row k (so its DPP row rg_k = k & 3 and register k >> 2 are
// static): the pivot column is the largest live entry of the WHOLE row (all columns sit in one 16-lane DPP row: true
// partial pivoting, no fallback needed), the scaled pivot row crosses to the other DPP rows by ds_bpermute, the
// multipliers come from lane (rg, pl) by the same 64-bit DPP row_newbcast / computed jump as in the big kernel, and the
// sign of the column permutation is accumulated from the live-column mask.
template <bool STEP>
__global__ __launch_bounds__(256) void hk_step_w16_kernel(StepArgs A) {
    __shared__ double prop[4][4][16];
    __shared__ double wsum[4];
    const int D = A.st.dim, DD = D * D, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rg = lane >> 4, tj = lane & 15;
    const double dt = A.dt, hh = 0.5 * dt, h6 = dt / 6.0;
    const bool colok = tj < D;
    const double sib = colok ? A.hk.si[tj] : 1.0, isib = 1.0 / sib;
    double sta[4], ista[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int r = rg + 4 * s;
        sta[s] = r < D ? A.hk.st[r] : 1.0;
        ista[s] = 1.0 / sta[s];
    }
    double esum = 0.0;
    for (int64_t tr = (int64_t)blockIdx.x * 4 + wave; tr < A.st.n; tr += (int64_t)gridDim.x * 4) {
        double *qp = A.st.qp + tr * 2 * D;
        double *M = A.st.mono + tr * 4 * (int64_t)DD;
        double p11[4] = {1, 1, 1, 1}, p12[4] = {0, 0, 0, 0}, p21[4] = {0, 0, 0, 0}, p22[4] = {1, 1, 1, 1};
        if (STEP) {
            double red5[5] = {0, 0, 0, 0, 0};
            if (lane < D) {
                const double q = qp[lane], p = qp[D + lane], im = A.pot.inv_mass[lane];
                const double c0 = A.pot.par0[lane], c1 = A.pot.par1 ? A.pot.par1[lane] : 0.0;
                double v, g, h1, h2, h3, h4;
                sep_eval(A.pot.kind, c0, c1, q, v, g, h1);
                const double kq1 = p * im, kp1 = -g;
                red5[0] = 0.5 * p * p * im - v;
                const double q2 = q + hh * kq1, p2 = p + hh * kp1;
                sep_eval(A.pot.kind, c0, c1, q2, v, g, h2);
                const double kq2 = p2 * im, kp2 = -g;
                red5[1] = 0.5 * p2 * p2 * im - v;
                const double q3 = q + hh * kq2, p3 = p + hh * kp2;
                sep_eval(A.pot.kind, c0, c1, q3, v, g, h3);
                const double kq3 = p3 * im, kp3 = -g;
                red5[2] = 0.5 * p3 * p3 * im - v;
                const double q4 = q + dt * kq3, p4 = p + dt * kp3;
                sep_eval(A.pot.kind, c0, c1, q4, v, g, h4);
                const double kq4 = p4 * im, kp4 = -g;
                red5[3] = 0.5 * p4 * p4 * im - v;
                red5[4] = 0.5 * p4 * p4 * im + v;
                qp[lane] = q + h6 * (kq1 + 2.0 * kq2 + 2.0 * kq3 + kq4);
                qp[D + lane] = p + h6 * (kp1 + 2.0 * kp2 + 2.0 * kp3 + kp4);
                double u1 = 1.0, v1 = 0.0, u2 = 0.0, v2 = 1.0;
                rk4_pair(u1, v1, im, h1, h2, h3, h4, dt);
                rk4_pair(u2, v2, im, h1, h2, h3, h4, dt);
                prop[wave][0][lane] = u1; prop[wave][1][lane] = u2; prop[wave][2][lane] = v1; prop[wave][3][lane] = v2;
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) red5[i] = wave_sum(red5[i]);
            if (lane == 0) {
                A.st.act[tr] += h6 * (red5[0] + 2.0 * red5[1] + 2.0 * red5[2] + red5[3]);
                esum += red5[4];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int r = rg + 4 * s;
                if (r < D) { p11[s] = prop[wave][0][r]; p12[s] = prop[wave][1][r]; p21[s] = prop[wave][2][r]; p22[s] = prop[wave][3][r]; }
            }
        }
        // ---- phase B
        cplx m[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int r = rg + 4 * s;
            const bool ok = colok && r < D;
            const int e = r * D + tj;
            double mqq = ok ? M[e] : 0.0, mqp = ok ? M[DD + e] : 0.0, mpq = ok ? M[2 * DD + e] : 0.0, mpp = ok ? M[3 * DD + e] : 0.0;
            if (STEP) {
                const double nqq = fma(p12[s], mpq, p11[s] * mqq), npq = fma(p22[s], mpq, p21[s] * mqq);
                const double nqp = fma(p12[s], mpp, p11[s] * mqp), npp = fma(p22[s], mpp, p21[s] * mqp);
                mqq = nqq; mpq = npq; mqp = nqp; mpp = npp;
                if (ok) { M[e] = mqq; M[DD + e] = mqp; M[2 * DD + e] = mpq; M[3 * DD + e] = mpp; }
            }
            m[s] = ok ? c_make(0.5 * (sta[s] * isib * mqq + ista[s] * sib * mpp),
                               0.5 * (-SC_HBAR * sta[s] * sib * mqp + (1.0 / SC_HBAR) * ista[s] * isib * mpq))
                      : c_make(r == tj ? 1.0 : 0.0, 0.0);
        }
        // ---- phase C
        cplx det = c_make(1.0, 0.0);
        bool live = colok, singular = false;
        int parity = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k < D && !singular) {
                constexpr int dummy = 0; (void)dummy;
                const int rgk = k & 3, sk = k >> 2;
                const int blk = (__double2hiint(c_abs2(m[sk])) & ~15) | (15 - tj);
                const int keymax = row16_max_i32((live && rg == rgk) ? blk : -1);
                const int kb = __builtin_amdgcn_readlane(keymax, 16 * rgk);
                const int pl = 15 - (kb & 15);
                const int src = 16 * rgk + pl;
                const cplx piv = c_make(readlane_f64(m[sk].x, src), readlane_f64(m[sk].y, src));
                if (piv.x == 0.0 && piv.y == 0.0) { singular = true; }
                else {
                    det = c_mul(det, piv);
                    const cplx inv = c_inv_fast(piv);
                    // sign of the column permutation: live columns to the left of the pivot column
                    const unsigned long long lm = __ballot(live && rg == 0);
                    parity ^= __popcll(lm & ((1ull << pl) - 1ull)) & 1;
                    // scaled pivot row at this lane's column (from DPP row rgk)
                    const cplx rowv = c_make(__shfl(m[sk].x, 16 * rgk + tj, 64), __shfl(m[sk].y, 16 * rgk + tj, 64));
                    const bool keep = live && tj != pl;
                    const cplx rs = c_mul(rowv, inv);
                    const cplx r = c_make(keep ? rs.x : 0.0, keep ? rs.y : 0.0);
                    cplx c[4];
                    column_fetch_n(pl, m[0], m[1], m[2], m[3], c[0], c[1], c[2], c[3]);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        if (rg + 4 * s <= k) c[s] = c_make(0.0, 0.0);        // rows 0..k are finished
                        m[s] = c_fnma(c[s], r, m[s]);
                    }
                    live = live && tj != pl;
                }
            }
        }
        if (lane == 0) {
            if (singular) det = c_make(0.0, 0.0);
            else if (parity) det = c_make(-det.x, -det.y);
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
    if (lane == 0) wsum[wave] = esum;
    __syncthreads();
    if (threadIdx.x == 0 && A.epart && STEP) A.epart[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

}  // namespace

#ifdef SC_TUNING
int sc_launch_step_ws(const StepArgs &a, hipStream_t s);       // tools/variants/sc_hk_step_ws.hip (measured, not adopted)
#endif

// launch the fast path; the caller has validated the arguments (separable potential, diag prefactor, D <= 64)
int sc_launch_step_sd(const StepArgs &a, hipStream_t s) {
    const int D = a.st.dim, nr = (D + 15) / 16, grid = sc_step_grid(a.st.n, D);
    int occ = 4;
    bool wave_kernel = D <= 16;
#ifdef SC_TUNING
    if (const char *occ_env = getenv("SC_SD_OCC")) occ = atoi(occ_env);   // waves per SIMD the NR=4 kernel is compiled for
    if (getenv("SC_NO_WAVE_KERNEL")) wave_kernel = false;
#endif
    if (wave_kernel) {
        bool packed = true;
#ifdef SC_TUNING
        if (getenv("SC_NO_SEP16")) packed = false;
#endif
        if (packed) {
            const int rc = sc_launch_step_sep16(a, grid, s);      // four trajectories per wavefront
            if (rc != 0) return rc < 0 ? rc : SC_OK;
        }
        // partial sums: only the first `wg` entries are written, the energy guard adds sc_step_grid() of them
        const int64_t quads = (a.st.n + 3) / 4;
        const int wg = (int)(quads < 2048 ? quads : 2048);
        if (a.epart && hipMemsetAsync(a.epart, 0, sizeof(double) * (size_t)grid, s) != hipSuccess)
            return sc_check_launch("sc_hk_step (partials)");
        if ((a.mode & 0xff) == 0) hipLaunchKernelGGL(hk_step_w16_kernel<true>, dim3(wg), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(hk_step_w16_kernel<false>, dim3(wg), dim3(256), 0, s, a);
        return sc_check_launch("sc_hk_step (one wavefront per trajectory)");
    }
    if ((a.mode & 0xff) == 0) hipLaunchKernelGGL(hk_modes_kernel, dim3(grid), dim3(256), 0, s, a);
    const bool step = (a.mode & 0xff) == 0;
    const bool tiled = a.st.mono_layout == SC_MONO_TILED16;
#ifdef SC_TUNING
    // SC_WS=1: the wave-specialised schedule (one producer + three eliminator groups per CU).  Measured on MI355X at
    // n = 1e5, D = 60: 6.71 ms against 5.98 ms of this kernel -- the eliminations are bound by VALU issue, not by their
    // latency, so giving them a CU of their own does not help (docs/NOTEBOOK.md section 8).
    if (nr == 4 && step && a.mode == 0 && getenv("SC_WS")) return sc_launch_step_ws(a, s);
#endif
    int sdgrid = grid;         // the energy partials belong to hk_modes_kernel: this kernel's grid is free
#ifdef SC_TUNING
    if (const char *g = getenv("SC_SD_GRID")) sdgrid = (int)std::min<int64_t>(a.st.n, atoi(g));
#endif
#define SC_LAUNCH_SD(NR_, OCC_)                                                                                         \
    do {                                                                                                                \
        if (step && tiled) hipLaunchKernelGGL((hk_step_sd_kernel<NR_, OCC_, true, true>), dim3(sdgrid), dim3(256), 0, s, a, MultiArgs{});   \
        else if (step) hipLaunchKernelGGL((hk_step_sd_kernel<NR_, OCC_, true, false>), dim3(sdgrid), dim3(256), 0, s, a, MultiArgs{});     \
        else if (tiled) hipLaunchKernelGGL((hk_step_sd_kernel<NR_, OCC_, false, true>), dim3(sdgrid), dim3(256), 0, s, a, MultiArgs{});    \
        else hipLaunchKernelGGL((hk_step_sd_kernel<NR_, OCC_, false, false>), dim3(sdgrid), dim3(256), 0, s, a, MultiArgs{});              \
    } while (0)
    switch (nr) {
        case 1: SC_LAUNCH_SD(1, 4); break;
        case 2: SC_LAUNCH_SD(2, 4); break;
        case 3: SC_LAUNCH_SD(3, 4); break;
        default:
#ifdef SC_TUNING
            if (occ == 3) { SC_LAUNCH_SD(4, 3); break; }
            if (occ < 3) { SC_LAUNCH_SD(4, 2); break; }
#endif
            (void)occ;
            SC_LAUNCH_SD(4, 4);
            break;
    }
#undef SC_LAUNCH_SD
    return sc_check_launch("sc_hk_step (separable/diagonal fast path)");
}

// TWO time steps per visit (sc_hk_step_multi; the caller has validated: separable potential, diagonal widths, 16 < D <= 64, tiled
// blocks, flags present): hk_modes_multi_kernel<2>, then the block kernel with two sub-steps per trajectory.
int sc_launch_step_sd_multi(const StepArgs &a, const sc_multi_scratch &ms, hipStream_t s) {
    const int D = a.st.dim, nr = (D + 15) / 16, grid = sc_step_grid(a.st.n, D);
    ModesMultiArgs mm{ms.work, ms.qp_mid, ms.act_mid};
    hipLaunchKernelGGL(hk_modes_multi_kernel<2>, dim3(grid), dim3(256), 0, s, a, mm);
    int rc = sc_check_launch("sc_hk_step_multi (modes)");
    if (rc) return rc;
    MultiArgs ma{ms.work, ms.c2_mid, ms.sgn_mid, ms.unrepaired};
    switch (nr) {
        case 2: hipLaunchKernelGGL((hk_step_sd_kernel<2, 4, true, true, 2>), dim3(grid), dim3(256), 0, s, a, ma); break;
        case 3: hipLaunchKernelGGL((hk_step_sd_kernel<3, 4, true, true, 2>), dim3(grid), dim3(256), 0, s, a, ma); break;
        case 4:
#ifdef SC_TUNING
            if (const char *occ_env = getenv("SC_SD_OCC")) {
                if (atoi(occ_env) == 3) { hipLaunchKernelGGL((hk_step_sd_kernel<4, 3, true, true, 2>), dim3(grid), dim3(256), 0, s, a, ma); break; }
                if (atoi(occ_env) == 2) { hipLaunchKernelGGL((hk_step_sd_kernel<4, 2, true, true, 2>), dim3(grid), dim3(256), 0, s, a, ma); break; }
            }
#endif
            hipLaunchKernelGGL((hk_step_sd_kernel<4, 4, true, true, 2>), dim3(grid), dim3(256), 0, s, a, ma); break;
        default: return sc_fail(SC_ERR_UNSUPPORTED, "sc_hk_step_multi: D = %d", D);
    }
    return sc_check_launch("sc_hk_step_multi (two steps per visit)");
}

#ifdef LU_PIVOT_CLOCK
extern "C" int sc_lu_pivot_clock(unsigned long long *buf) {       // device buffer of 64 * 64 * 4 time stamps, or NULL to switch off
    return hipMemcpyToSymbol(HIP_SYMBOL(g_lu_clock), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#endif
