// In-place change of the storage order of the monodromy blocks (include/semiclassical_hip.h: SC_MONO_ROWMAJOR <->
// SC_MONO_TILED16).  One workgroup per trajectory: the 4 D^2 doubles of a trajectory (at most 128 KB at D = 64) are
// read linearly into LDS and written back in the other order.  Off the hot path: the propagator converts when a
// state enters or leaves the separable fast path of sc_hk_step.
#include "sc_common.h"


namespace {

__global__ __launch_bounds__(256) void mono_convert_kernel(double *mono, int64_t n, int D, int from, int to) {
    extern __shared__ double2 smem2[];
    double *buf = (double *)smem2;
    const int DD = D * D, tid = threadIdx.x;
    for (int64_t tr = blockIdx.x; tr < n; tr += gridDim.x) {
        double *M = mono + tr * 4 * (int64_t)DD;
        __syncthreads();
        for (int e = tid; e < 4 * DD; e += 256) buf[e] = M[e];
        __syncthreads();
        for (int e = tid; e < 4 * DD; e += 256) {
            const int p = e / DD, ab = e - p * DD, a = ab / D, b = ab - a * D;
            M[sc_mono_offset(to, D, p, a, b)] = buf[sc_mono_offset(from, D, p, a, b)];
        }
    }
}

// mono[i][p] <- left[p] . mono[i][p] . right[p] for the four blocks p of every trajectory (D <= 16, row-major): the change of basis
// around sc_hk_run_modal.  One workgroup per trajectory at a time; constants and the blocks in LDS.
__global__ __launch_bounds__(256) void mono_similarity_kernel(double *mono, int64_t n, int D, const double *left, const double *right) {
    __shared__ double sL[4 * 256], sR[4 * 256], sM[4 * 256], sT[4 * 256];
    const int DD = D * D, tid = threadIdx.x;
    for (int e = tid; e < 4 * DD; e += 256) { sL[e] = left[e]; sR[e] = right[e]; }
    for (int64_t tr = blockIdx.x; tr < n; tr += gridDim.x) {
        double *M = mono + tr * 4 * (int64_t)DD;
        __syncthreads();
        for (int e = tid; e < 4 * DD; e += 256) sM[e] = M[e];
        __syncthreads();
        for (int e = tid; e < 4 * DD; e += 256) {
            const int p = e / DD, ij = e - p * DD, i = ij / D, j = ij - i * D;
            double acc = 0.0;
            for (int k = 0; k < D; ++k) acc = fma(sM[p * DD + i * D + k], sR[p * DD + k * D + j], acc);
            sT[e] = acc;
        }
        __syncthreads();
        for (int e = tid; e < 4 * DD; e += 256) {
            const int p = e / DD, ij = e - p * DD, i = ij / D, j = ij - i * D;
            double acc = 0.0;
            for (int k = 0; k < D; ++k) acc = fma(sL[p * DD + i * D + k], sT[p * DD + k * D + j], acc);
            M[e] = acc;
        }
    }
}

}  // namespace

extern "C" int sc_mono_similarity(const sc_state *st, const double *left, const double *right, void *stream) {
    if (!st || !st->mono || !left || !right) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_mono_similarity: null argument");
    if (st->dim < 1 || st->dim > 16) return sc_fail(SC_ERR_UNSUPPORTED, "sc_mono_similarity: D=%d outside 1..16", st->dim);
    if (int rq = sc_require_rowmajor(st, "sc_mono_similarity")) return rq;
    if (st->n <= 0) return SC_OK;
    const int grid = (int)(st->n < 8192 ? st->n : 8192);
    hipLaunchKernelGGL(mono_similarity_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, st->mono, st->n, st->dim, left, right);
    return sc_check_launch("sc_mono_similarity");
}

extern "C" int sc_mono_convert(const sc_state *st, int32_t to_layout, void *stream) {
    if (!st || !st->mono) return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_mono_convert: null argument");
    const int from = st->mono_layout, D = st->dim;
    if ((from != SC_MONO_ROWMAJOR && from != SC_MONO_TILED16) || (to_layout != SC_MONO_ROWMAJOR && to_layout != SC_MONO_TILED16))
        return sc_fail(SC_ERR_BAD_ARGUMENT, "sc_mono_convert: unknown layout %d -> %d", from, to_layout);
    if (D < 1 || D > 64) return sc_fail(SC_ERR_UNSUPPORTED, "sc_mono_convert: D=%d outside 1..64", D);
    if (from == to_layout || D <= 16 || st->n <= 0) return SC_OK;          // the two orders coincide for D <= 16
    const size_t lds = (size_t)4 * D * D * sizeof(double);
    if (hipFuncSetAttribute((const void *)mono_convert_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return sc_check_launch("sc_mono_convert (LDS attribute)");
    const int grid = (int)(st->n < 2048 ? st->n : 2048);
    hipLaunchKernelGGL(mono_convert_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, st->mono, st->n, D, from, (int)to_layout);
    return sc_check_launch("sc_mono_convert");
}

// entry points that read st->mono in the row-major order call this first
int sc_require_rowmajor(const sc_state *st, const char *who) {
    if (st->mono_layout != SC_MONO_ROWMAJOR && st->dim > 16)
        return sc_fail(SC_ERR_BAD_ARGUMENT, "%s: the state is in the tiled monodromy layout (sc_mono_convert it first)", who);
    return SC_OK;
}
