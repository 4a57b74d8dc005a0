// Micro-benchmark (GPU box): does the memory-side cache absorb a SECOND visit of a trajectory's monodromy blocks?
//
// hk_step_sd_kernel reads and writes the 4 D^2 doubles of every trajectory once per time step and runs at the rate of a plain copy
// (DESIGN.md section 7.1).  A kernel that advances a trajectory by K time steps per visit -- store the blocks, eliminate, read them
// back, apply the next step's row propagators, ... -- touches HBM once per K steps only IF the intermediate stores and reloads are
// served by the L2 / the 256 MB memory-side cache: with 1024 workgroups in flight 118 MB of blocks are "between two sub-steps".
// Here: the same bytes, the same launch shape (1024 persistent workgroups x 256 threads, grid-stride over the trajectories, 16 B per
// lane, linear order), K read-modify-write passes over a trajectory back to back before the workgroup moves on -- time per pass for
// K = 1, 2, 4, with plain and with non-temporal accesses, and with a delay between the passes (the elimination takes ~20 us).
// build: hipcc --offload-arch=gfx950 -O3 -o revisit revisit.hip ; run: ./revisit [n] [D]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <int K, bool NT, int DELAY>
__global__ __launch_bounds__(256, 4) void revisit(double *mono, long n, long units) {      // units: 16-byte units per trajectory
    for (long tr = blockIdx.x; tr < n; tr += gridDim.x) {
        d2 *M = (d2 *)mono + tr * units;
#pragma unroll 1
        for (int k = 0; k < K; ++k) {
            for (long c0 = 0; c0 < units; c0 += 256 * 8) {
                d2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const long c = c0 + u * 256 + threadIdx.x;
                    v[u] = c < units ? (NT ? __builtin_nontemporal_load(M + c) : M[c]) : (d2){0.0, 0.0};
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const long c = c0 + u * 256 + threadIdx.x;
                    const d2 o = {v[u].x * 1.0000001 + v[u].y * 1e-9, v[u].y * 1.0000001 - v[u].x * 1e-9};
                    if (c < units) { if (NT) __builtin_nontemporal_store(o, M + c); else M[c] = o; }
                }
            }
            if (DELAY > 0 && k + 1 < K) {                       // the elimination between two sub-steps
                const long long t0 = wall_clock64();
                while (wall_clock64() - t0 < DELAY) __builtin_amdgcn_s_sleep(8);      // 100 MHz ticks
            }
        }
    }
}

template <int K, bool NT, int DELAY>
static double run(double *mono, long n, long units, const char *what) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((revisit<K, NT, DELAY>), dim3(1024), dim3(256), 0, 0, mono, n, units);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = 2.0 * 16.0 * units * n;          // read + write of one pass
    printf("%-44s K = %d : %8.3f ms per launch  %7.3f ms per pass  (%5.2f TB/s per pass; one pass from HBM both ways = %.2f GB)\n", what, K,
           best, best / K, bytes / (best / K * 1e-3) / 1e12, bytes / 1e9);
    return best;
}

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 100000;
    const int D = argc > 2 ? atoi(argv[2]) : 60;
    const long units = 4L * D * D / 2;
    double *mono;
    CHECK(hipMalloc(&mono, sizeof(double) * 2 * units * n));
    CHECK(hipMemset(mono, 0, sizeof(double) * 2 * units * n));
    // correctness of the re-read: every unit starts as (1, 2); K passes of the rotation-like update have a known result.  A stale
    // line in the vector L1 (the second pass re-reading what the first pass of the same thread stored) would show here.
    {
        const long nchk = 4096;          // trajectories
        d2 *h = (d2 *)malloc(sizeof(d2) * units * nchk);
        for (long i = 0; i < units * nchk; ++i) h[i] = (d2){1.0, 2.0};
        CHECK(hipMemcpy(mono, h, sizeof(d2) * units * nchk, hipMemcpyHostToDevice));
        hipLaunchKernelGGL((revisit<4, false, 0>), dim3(1024), dim3(256), 0, 0, mono, nchk, units);
        CHECK(hipMemcpy(h, mono, sizeof(d2) * units * nchk, hipMemcpyDeviceToHost));
        double x = 1.0, y = 2.0;
        for (int k = 0; k < 4; ++k) { const double nx = x * 1.0000001 + y * 1e-9, ny = y * 1.0000001 - x * 1e-9; x = nx; y = ny; }
        long bad = 0;
        for (long i = 0; i < units * nchk; ++i) bad += (h[i].x != x || h[i].y != y);
        printf("# 4 plain passes over %ld trajectories: %ld of %ld units differ from the expected value (%s)\n", nchk, bad, units * nchk,
               bad ? "STALE READS" : "every re-read saw the previous pass's store");
        free(h);
        CHECK(hipMemset(mono, 0, sizeof(double) * 2 * units * n));
    }
    printf("# n = %ld trajectories, D = %d: %.1f KB per trajectory, %.2f GB; 1024 workgroups in flight hold %.0f MB\n", n, D,
           16.0 * units / 1e3, 16.0 * units * n / 1e9, 1024 * 16.0 * units / 1e6);
    const double t1 = run<1, false, 0>(mono, n, units, "plain loads / stores");
    run<2, false, 0>(mono, n, units, "plain, passes back to back");
    run<4, false, 0>(mono, n, units, "plain, passes back to back");
    run<2, false, 2000>(mono, n, units, "plain, 20 us between the passes");
    run<4, false, 2000>(mono, n, units, "plain, 20 us between the passes");
    run<1, true, 0>(mono, n, units, "non-temporal loads / stores");
    run<2, true, 0>(mono, n, units, "non-temporal, passes back to back");
    run<2, true, 2000>(mono, n, units, "non-temporal, 20 us between the passes");
    (void)t1;
    return 0;
}
