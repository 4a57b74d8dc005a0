// Micro-benchmark (GPU box): does the DP-ALU DPP form  v_fmac_f64_dpp ... row_newbcast:K  (the only 64-bit VALU op that
// takes a DPP operand on gfx90a+/gfx950) compute  acc += b[lane K of the 16-lane row] * a  correctly and at the rate of a
// plain v_fmac_f64?  It replaces the pair  v_mov_b64_dpp (broadcast) + v_fmac_f64  in the small-matrix kernels.
// build: hipcc --offload-arch=gfx950 -O3 -o dpp_fmac dpp_fmac.hip ; run: ./dpp_fmac
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>   // 0: plain fmac, 1: fused DPP fmac, 2: mov_dpp + fmac
__global__ __launch_bounds__(256) void chain(double *out, int iters) {
    const int lane = threadIdx.x & 63;
    double acc[16], b[4];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = 1.0 + lane + 100.0 * j;          // differs per lane: the broadcast is visible
    double a = 1e-3 * (1 + (threadIdx.x >> 6));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0)
                asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[j]) : "v"(b[j & 3]), "v"(a));
            else if (MODE == 1) {
                if ((j & 3) == 0) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(acc[j]) : "v"(b[j & 3]), "v"(a));
                if ((j & 3) == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc[j]) : "v"(b[j & 3]), "v"(a));
                if ((j & 3) == 2) asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:11 row_mask:0xf bank_mask:0xf" : "+v"(acc[j]) : "v"(b[j & 3]), "v"(a));
                if ((j & 3) == 3) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "+v"(acc[j]) : "v"(b[j & 3]), "v"(a));
            } else {
                double t;
                if ((j & 3) == 0) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(b[j & 3]));
                if ((j & 3) == 1) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(b[j & 3]));
                if ((j & 3) == 2) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:11 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(b[j & 3]));
                if ((j & 3) == 3) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(b[j & 3]));
                asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[j]) : "v"(t), "v"((j & 3) == 2 ? -a : a));
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[j] * (1 + j);
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int grid = 1024 * 2, iters = 20000;      // 2 workgroups of 4 waves per CU: 2 waves per SIMD
    double *out, *h = (double *)malloc(sizeof(double) * grid * 256);
    CHECK(hipMalloc(&out, sizeof(double) * grid * 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double ref[3][256];
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(grid), dim3(256), 0, 0, out, iters);
            if (mode == 1) hipLaunchKernelGGL(chain<1>, dim3(grid), dim3(256), 0, 0, out, iters);
            if (mode == 2) hipLaunchKernelGGL(chain<2>, dim3(grid), dim3(256), 0, 0, out, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
        }
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h, out, sizeof(double) * 256, hipMemcpyDeviceToHost));
        for (int t = 0; t < 256; ++t) ref[mode][t] = h[t];
        // 16 fmac per iteration per wave; waves per SIMD = grid * 4 / 1024
        const double wave_instr = 16.0 * iters, waves_per_simd = grid * 4.0 / 1024.0;
        const double cyc = ms * 1e-3 * 2.4e9 / (wave_instr * waves_per_simd);
        printf("%s: %.3f ms  -> %.2f cycles (at 2.4 GHz) per fmac per SIMD  (%.1f TFLOP/s)\n",
               mode == 0 ? "plain v_fmac_f64        " : mode == 1 ? "v_fmac_f64_dpp newbcast " : "v_mov_b64_dpp + v_fmac  ",
               ms, cyc, 128.0 * wave_instr * grid * 4 / (ms * 1e-3) / 1e12);
    }
    // expected value for thread t (mode 1 and 2): acc_j = iters * a * sign_j * b_j[lane K_j of the row]
    double maxerr = 0.0;
    const int K[4] = {0, 5, 11, 15};
    for (int t = 0; t < 256; ++t) {
        const int lane = t & 63, row = lane & ~15;
        const double a = 1e-3 * (1 + (t >> 6));
        double s = 0.0;
        for (int j = 0; j < 16; ++j) {
            const double b = 1.0 + (row + K[j & 3]) + 100.0 * (j & 3);
            s += (double)iters * a * ((j & 3) == 2 ? -1.0 : 1.0) * b * (1 + j);
        }
        maxerr = fmax(maxerr, fabs(ref[1][t] - s) / fabs(s));
        maxerr = fmax(maxerr, fabs(ref[2][t] - ref[1][t]) / fabs(s));
    }
    printf("fused DPP result vs expected broadcast semantics / vs mov+fmac: max rel. deviation %.2e\n", maxerr);
    return maxerr < 1e-9 ? 0 : 1;
}
