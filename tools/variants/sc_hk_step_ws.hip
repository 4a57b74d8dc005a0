// Wave-specialised variant of the separable fast path for 48 < D <= 64 (the headline workload, D = 60).
//
// hk_step_sd_kernel gives a trajectory ONE 256-thread workgroup that alternates between streaming its monodromy
// blocks (phase B: HBM bound) and eliminating the prefactor matrix (phase C: a chain of 60 dependent pivots, ~33 us).
// With four such workgroups on a compute unit about half of them sit in the pivot chain at any time and the other
// half cannot keep the memory system busy: 6.1 ms per step where streaming alone takes 4.5 ms and the eliminations
// alone 3.2 ms (tools/phase_timing.py).
//
// Here a compute unit runs ONE workgroup of 16 wavefronts with fixed roles:
//   wavefronts 0..3    "producer": phase B of trajectory after trajectory, without ever eliminating.  No matrix lives in
//                      its registers, so two 16-row slots of loads are in flight per thread (the 128-VGPR budget of the
//                      old kernel allowed one).  The prefactor matrix goes to an LDS slot in the register layout of the
//                      eliminating threads.
//   wavefronts 4..15   three "eliminator" groups of four wavefronts.  Group g takes every third matrix from the slot
//                      into registers and runs the block-pivoted elimination of sc_hk_lu.h on it, with its own row
//                      buffers, pivot records and a four-wavefront barrier built on an LDS counter.
// Every SIMD hosts one producer wavefront and one wavefront of each eliminator group.  Hand-over is per wavefront
// (producer wavefront w writes exactly the elements eliminator wavefront w of the target group holds), by a tag in LDS:
// full[w] = k + 1 when the k-th matrix of the workgroup is in place, 0 when it has been taken.  All waits are polls
// with an iteration cap: a protocol error ends in NaN prefactors, never in a hung GPU.
//
// Same arithmetic, element for element, as hk_step_sd_kernel<4, 4, true, TILED> (reference propagators.py:86-119,
// 313-383, 951-1052); parity-tested (tests/test_hk_gpu.py ran green with it).
//
// MEASURED AND NOT ADOPTED (MI355X, n = 1e5, D = 60, one A/B run): 6.71 ms per step launch against 5.98 ms of
// hk_step_sd_kernel; with the eliminations switched off the producer alone streams the step in 4.68 ms (5.0 TB/s).
// Three eliminator groups deliver one determinant per ~17 us per CU -- the same rate as the four alternating workgroups
// of the adopted kernel: the elimination costs ~36 000 wavefront instructions per trajectory (~90 per pivot and
// wavefront, a third of them the multiply-adds) and the CU's VALU issue slots are what it is bound by.  Compiled only
// into the tuning build (tools/mkvar.sh, SC_WS=1).
#include "sc_common.h"
#include "sc_hk_lu.h"

namespace {

constexpr int WS_GROUPS = 3;
constexpr int WS_SPIN_LIMIT = 1 << 22;

struct WsLds {
    double scl[4 * 64];                       // st, 1/st, si, 1/si
    cplx rowbuf[WS_GROUPS][16][64];
    PivotRecord pivrec[WS_GROUPS][16];
    int weak[WS_GROUPS];
    cplx detbuf[WS_GROUPS][16];
    int bar[WS_GROUPS];                       // monotonic arrival counters of the group barriers
    int full[4];                              // hand-over tags, one per wavefront position
    int err;
    int pad[3];
    cplx slot[16][256];                       // prefactor matrix in flight: [ra * 4 + rb][thread of the group]
};

__device__ __forceinline__ int lds_load(const int *p) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}

// wait until *p == want (wave-uniform); false after WS_SPIN_LIMIT polls
__device__ __forceinline__ bool wait_for(const int *p, int want) {
    for (int spins = 0; spins < WS_SPIN_LIMIT; ++spins) {
        if (lds_load(p) == want) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// barrier of the four wavefronts of an eliminator group: every wavefront adds one to the group's counter and waits
// until all four arrivals of this round are in.  LDS operations of a wavefront execute in issue order, so what a
// wavefront wrote before the barrier is visible to the others after it.
__device__ __forceinline__ void group_barrier(int *counter, int &target, int lane, int *err) {
    target += 4;
    __asm__ volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    bool ok = false;
    for (int spins = 0; spins < WS_SPIN_LIMIT; ++spins) {
        if (lds_load(counter) - target >= 0) { ok = true; break; }
    }
    if (!ok && lane == 0) *err = 1;
    __asm__ volatile("" ::: "memory");
}

template <bool TILED>
__global__ __launch_bounds__(1024, 1) void hk_step_ws_kernel(StepArgs A) {
    extern __shared__ double2 smem2[];
    WsLds &S = *reinterpret_cast<WsLds *>(smem2);
    constexpr int NR = 4;
    const int D = A.st.dim, DD = D * D, tid = threadIdx.x, ltid = tid & 255, role = tid >> 8, lane = tid & 63, w = ltid >> 6;
    const int tj = ltid & 15, trow = w * 4 + ((ltid >> 4) & 3);
    constexpr int NCL_BASE = 16 * (NR - 1);
    const unsigned toff = TILED ? (unsigned)(trow * 16 + tj) : (unsigned)(trow * D + tj);
    const unsigned toffl = (unsigned)(trow * (D - NCL_BASE) + tj);

    if (tid < 64) {
        const bool in = tid < D;
        const double st = in ? A.hk.st[tid] : 1.0, si = in ? A.hk.si[tid] : 1.0;
        S.scl[tid] = st; S.scl[64 + tid] = 1.0 / st; S.scl[128 + tid] = si; S.scl[192 + tid] = 1.0 / si;
        if (tid < 16) for (int g = 0; g < WS_GROUPS; ++g) S.pivrec[g][tid].pad = 0;
        if (tid < WS_GROUPS) { S.weak[tid] = 0; S.bar[tid] = 0; }
        if (tid < 4) S.full[tid] = 0;
        if (tid == 0) S.err = 0;
    }
    __syncthreads();          // the only workgroup-wide barrier

    const int64_t n = A.st.n;
    if (role == 0) {
        // =============================== producer: phase B of every trajectory of this workgroup ===============================
        // The 16-row slots of consecutive trajectories form ONE stream: two slots of loads are in flight at any time, also
        // across the boundary between two trajectories (a lone producer has no other workgroup to hide that bubble).
        double v[2][4][NR];
        auto load_slot = [&](const double *Mx, int ra, double (&buf)[4][NR]) {
            const bool rowok = 16 * ra + trow < D;
            const int nra = min(16, D - 16 * ra);
#pragma unroll
            for (int rb = 0; rb < NR; ++rb) {
                const bool ok = rowok && 16 * rb + tj < D;
                const int ncb = rb == NR - 1 ? D - NCL_BASE : 16;
                const double *pe = Mx + __builtin_amdgcn_readfirstlane(TILED ? 4 * (16 * ra * D + nra * 16 * rb) : 16 * ra * D + 16 * rb);
                const int plane = __builtin_amdgcn_readfirstlane(TILED ? nra * ncb : DD);
                const unsigned to = (TILED && rb == NR - 1) ? toffl : toff;
                buf[0][rb] = ok ? pe[to] : 0.0;
                buf[1][rb] = ok ? pe[plane + to] : 0.0;
                buf[2][rb] = ok ? pe[2 * plane + to] : 0.0;
                buf[3][rb] = ok ? pe[3 * plane + to] : 0.0;
            }
        };
        if ((int64_t)blockIdx.x < n) {
            const double *M0 = A.st.mono + (int64_t)blockIdx.x * 4 * (int64_t)DD;
            load_slot(M0, 0, v[0]);
            load_slot(M0, 1, v[1]);
        }
        int k = 0;
        for (int64_t tr = blockIdx.x; tr < n; tr += gridDim.x, ++k) {
            double *M = A.st.mono + tr * 4 * (int64_t)DD;
            const bool more = tr + gridDim.x < n;
            const double *Mn = A.st.mono + (more ? tr + gridDim.x : tr) * 4 * (int64_t)DD;
            const double *pr = A.st.work + tr * 4 * (int64_t)D;       // row propagators P_a from hk_modes_kernel
            int til = trow, tjl = tj;
            __asm__ volatile("" : "+v"(til), "+v"(tjl));            // see hk_step_sd_kernel: keeps LDS indices out of scratch
#pragma unroll
            for (int ra = 0; ra < NR; ++ra) {
                const int a = 16 * ra + til;
                const bool rowok = a < D;
                const int al = a & 63, ar = rowok ? a : 0;
                const double p11 = pr[ar], p12 = pr[D + ar], p21 = pr[2 * D + ar], p22 = pr[3 * D + ar];
                const double sta = S.scl[al], ista = S.scl[64 + al];
                const int nra = min(16, D - 16 * ra);
                if (ra == 0) {
                    // the previous matrix must have been taken before this one goes into the slot
                    if (!wait_for(&S.full[w], 0) && lane == 0) S.err = 1;
                }
#pragma unroll
                for (int rb = 0; rb < NR; ++rb) {
                    const int b = 16 * rb + tj;
                    const bool ok = rowok && b < D;
                    const int ncb = rb == NR - 1 ? D - NCL_BASE : 16;
                    double *pe = M + __builtin_amdgcn_readfirstlane(TILED ? 4 * (16 * ra * D + nra * 16 * rb) : 16 * ra * D + 16 * rb);
                    const int plane = __builtin_amdgcn_readfirstlane(TILED ? nra * ncb : DD);
                    const unsigned to = (TILED && rb == NR - 1) ? toffl : toff;
                    const double oqq = v[ra & 1][0][rb], oqp = v[ra & 1][1][rb], opq = v[ra & 1][2][rb], opp = v[ra & 1][3][rb];
                    const double mqq = fma(p12, opq, p11 * oqq), mpq = fma(p22, opq, p21 * oqq);
                    const double mqp = fma(p12, opp, p11 * oqp), mpp = fma(p22, opp, p21 * oqp);
                    if (ok) { pe[to] = mqq; pe[plane + to] = mqp; pe[2 * plane + to] = mpq; pe[3 * plane + to] = mpp; }
                    const int bl = (16 * rb + tjl) & 63;
                    const double sib = S.scl[128 + bl], isib = S.scl[192 + bl];
                    S.slot[ra * 4 + rb][ltid] = ok ? c_make(0.5 * (sta * isib * mqq + ista * sib * mpp),
                                                            0.5 * (-SC_HBAR * sta * sib * mqp + (1.0 / SC_HBAR) * ista * isib * mpq))
                                                   : c_make(0.0, 0.0);
                }
                // refill the buffer just consumed: slot ra + 2 of this trajectory, or slot ra - 2 of the next one
                if (ra + 2 < NR) load_slot(M, ra + 2, v[ra & 1]);
                else if (more) load_slot(Mn, ra + 2 - NR, v[ra & 1]);
            }
            // this wavefront's part of matrix k is in the slot (its LDS writes are ordered before the tag)
            __asm__ volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_store(&S.full[w], k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __asm__ volatile("" ::: "memory");
        }
        return;
    }

    // =============================== eliminator group g = role - 1 ===============================
    const int g = role - 1;
    cplx (*rowbuf)[64] = S.rowbuf[g];
    PivotRecord *pivrec = S.pivrec[g];
    int *weak = &S.weak[g];
    cplx *detbuf = S.detbuf[g];
    int bar_target = 0, seq0 = 0;
    auto barrier = [&] { group_barrier(&S.bar[g], bar_target, lane, &S.err); };
    int k = g;
    for (int64_t tr = blockIdx.x + (int64_t)g * gridDim.x; tr < n; tr += (int64_t)WS_GROUPS * gridDim.x, k += WS_GROUPS, seq0 += 4) {
        if (ltid == 0) *weak = 0;
        if (ltid < 16) detbuf[ltid] = c_make(1.0, 0.0);
        // take matrix k from the slot
        const bool arrived = wait_for(&S.full[w], k + 1);
        cplx m[NR][NR];
#pragma unroll
        for (int ra = 0; ra < NR; ++ra)
#pragma unroll
            for (int rb = 0; rb < NR; ++rb) m[ra][rb] = S.slot[ra * 4 + rb][ltid];
        __asm__ volatile("" ::: "memory");
        if (lane == 0) {
            if (!arrived) S.err = 1;
            __hip_atomic_store(&S.full[w], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // after the reads (issue order)
        }
        __asm__ volatile("" ::: "memory");

        cplx det;
#ifndef WS_ABLATE_LU
        eliminate_block<NR, 0, 64>(m, detbuf, D, seq0 + 1, rowbuf, pivrec, weak, ltid, barrier);
        eliminate_block<NR, 1, 64>(m, detbuf, D, seq0 + 2, rowbuf, pivrec, weak, ltid, barrier);
        eliminate_block<NR, 2, 64>(m, detbuf, D, seq0 + 3, rowbuf, pivrec, weak, ltid, barrier);
        eliminate_block<NR, 3, 64>(m, detbuf, D, seq0 + 4, rowbuf, pivrec, weak, ltid, barrier);
#else
        if (ltid == 0) detbuf[0] = m[0][0];
#endif
        barrier();
        if (ltid == 0) {
            cplx *c2 = (cplx *)A.st.c2;
            if (lds_load(&S.err)) {
                c2[tr] = c_make(__builtin_nan(""), __builtin_nan(""));         // protocol failure: never silently wrong
            } else if ((*weak & 1) && A.st.flags) {
                A.st.flags[tr] = 1;                  // c2 / sgn are left to the fully pivoted fallback
                atomicAdd(&A.st.flags[A.st.n], 1);
            } else {
                det = (*weak & 2) ? c_make(0.0, 0.0) : finish_determinant(detbuf, row_order_is_odd(D));
                const cplx prev = c2[tr];
                if (prev.x < 0.0 && det.x < 0.0 && prev.y * det.y < 0.0) A.st.sgn[tr] = -A.st.sgn[tr];
                c2[tr] = det;
            }
        }
        barrier();            // permseq / weak are reused by the next trajectory
    }
}

}  // namespace

// launch for 48 < D <= 64, mode 0 (step + prefactor); the caller has run hk_modes_kernel in the same stream
int sc_launch_step_ws(const StepArgs &a, hipStream_t s) {
    const size_t lds = sizeof(WsLds);
    const bool tiled = a.st.mono_layout == SC_MONO_TILED16;
    const void *fn = tiled ? (const void *)hk_step_ws_kernel<true> : (const void *)hk_step_ws_kernel<false>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return sc_check_launch("sc_hk_step (wave-specialised kernel, LDS attribute)");
    const int64_t n = a.st.n;
    const int grid = (int)(n < 256 ? n : 256);           // one workgroup per compute unit
    if (tiled) hipLaunchKernelGGL(hk_step_ws_kernel<true>, dim3(grid), dim3(1024), lds, s, a);
    else hipLaunchKernelGGL(hk_step_ws_kernel<false>, dim3(grid), dim3(1024), lds, s, a);
    return sc_check_launch("sc_hk_step (wave-specialised kernel)");
}
