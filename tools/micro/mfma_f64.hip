// Micro-benchmark (GPU box): what does the FP64 matrix pipe of this MI355X sustain?
//
// Back-to-back v_mfma_f64_16x16x4_f64 (2048 flop per wave instruction) on NACC independent accumulator tiles per wave,
// 1 / 2 / 4 waves per SIMD on all 256 CUs, with the A/B operands
//   FEED 0  fixed registers (the bare pipe),
//   FEED 1  fetched per product from LDS, two doubles per lane (the compiler forms ds_read2_b64 / ds_read_b64 pairs):
//           the operand feed of dense_mono_mfma_slab_dma2 (csrc/sc_dense_mono.hip),
//   FEED 2  fixed registers, one independent v_fma_f64 issued per product (does the vector FP64 pipe share the matrix pipe?).
// Every kernel brackets its loop with s_memtime (shader clock) and s_memrealtime (100 MHz), so a line shows the cycles per
// instruction at the clock the part actually ran at next to the TFLOP/s from the HIP-event duration.  The datasheet's
// 78.6 TFLOP/s is one instruction per 64 cycles per SIMD at 2.4 GHz.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64 mfma_f64.hip ; run: ./mfma_f64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC, int FEED>
__global__ __launch_bounds__(256) void mfma_chain(double *out, unsigned long long *clk, int iters) {
    __shared__ double lds[4 * 16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 16 * 64; i += 256) lds[i] = 1e-3 * (1 + (i & 63));
    __syncthreads();
    v4d acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = v4d{0.0, 0.0, 0.0, 0.0};
    double a = 1e-3 * (1 + lane), b = 1.0 / (1 + lane);
    double v[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) v[j] = 1.0 + j;
    const double *row = lds + wave * 16 * 64 + lane;     // one double per lane and row: conflict-free
    const unsigned long long t0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        double fa[NACC], fb[NACC];
        if (FEED == 1) {                                           // all operand reads of the iteration first, then the products
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                const double *p = row + ((it + j) & 7) * 128;      // the address changes with the iteration: no hoisting
                fa[j] = p[0];
                fb[j] = p[64];
            }
        }
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            // inline assembly: with the builtin hipcc copies every accumulator VGPR <-> AGPR once per iteration
            if (FEED == 1) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(fa[j]), "v"(fb[j]));
            else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
            if (FEED == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[j]) : "v"(a), "v"(b));
        }
    }
    const unsigned long long t1 = clock64(), r1 = wall_clock64();
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3] + v[j];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {          // per workgroup: shader cycles of the loop, its begin and end on the 100 MHz clock
        clk[3 * blockIdx.x] = t1 - t0;
        clk[3 * blockIdx.x + 1] = r0;
        clk[3 * blockIdx.x + 2] = r1;
    }
}

template <int NACC, int FEED>
static void run(int waves_per_simd, double *out, unsigned long long *clk, const char *feed) {
    const int grid = 256 * waves_per_simd, iters = 40000 / NACC;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((mfma_chain<NACC, FEED>), dim3(grid), dim3(256), 0, 0, out, clk, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    static unsigned long long h[3 * 1024];
    CHECK(hipMemcpy(h, clk, sizeof(unsigned long long) * 3 * grid, hipMemcpyDeviceToHost));
    const double instr_per_wave = (double)iters * NACC;
    const double flops = 2048.0 * instr_per_wave * grid * 4;
    // how many workgroups were resident at a time: sum of the workgroups' loop durations over the span of the launch
    unsigned long long first = ~0ull, last = 0;
    double busy = 0.0, cycles = 0.0;
    for (int g = 0; g < grid; ++g) {
        if (h[3 * g + 1] < first) first = h[3 * g + 1];
        if (h[3 * g + 2] > last) last = h[3 * g + 2];
        busy += (double)(h[3 * g + 2] - h[3 * g + 1]);
        cycles += (double)h[3 * g];
    }
    const double resident = busy / (double)(last - first) / 256.0;                 // workgroups (= waves per SIMD) resident per CU
    const double mhz = 100.0 * cycles / busy;                                      // s_memrealtime ticks at 100 MHz
    const double cyc = cycles / grid / (instr_per_wave * resident);                // shader cycles per MFMA per SIMD
    printf("%-28s acc tiles %d  waves/SIMD %d (resident %.2f): %8.3f ms  %6.1f TFLOP/s  %6.1f shader cycles per MFMA per SIMD  clock %5.0f MHz\n",
           feed, NACC, waves_per_simd, resident, best, flops / (best * 1e-3) / 1e12, cyc, mhz);
}

int main() {
    double *out;
    unsigned long long *clk;
    CHECK(hipMalloc(&out, sizeof(double) * 256 * 4 * 256));
    CHECK(hipMalloc(&clk, 3 * 1024 * sizeof(unsigned long long)));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("# %s, %d CUs, clockRate %d kHz; v_mfma_f64_16x16x4_f64 = 2048 flop per wave instruction\n", prop.name,
           prop.multiProcessorCount, prop.clockRate);
    for (int w = 1; w <= 4; w *= 2) {
        run<1, 0>(w, out, clk, "registers");
        run<2, 0>(w, out, clk, "registers");
        run<4, 0>(w, out, clk, "registers");
        run<8, 0>(w, out, clk, "registers");
    }
    for (int w = 1; w <= 4; w *= 2) {
        run<4, 1>(w, out, clk, "LDS feed (2 doubles/lane)");
        run<8, 1>(w, out, clk, "LDS feed (2 doubles/lane)");
    }
    for (int w = 1; w <= 2; w *= 2) {
        run<4, 2>(w, out, clk, "registers + v_fma_f64 each");
        run<8, 2>(w, out, clk, "registers + v_fma_f64 each");
    }
    return 0;
}
