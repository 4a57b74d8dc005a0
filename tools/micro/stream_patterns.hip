// Micro-benchmark (GPU box): what does the in-place read-modify-write of one trajectory's monodromy blocks cost under
// different thread -> address mappings?  Same launch shape as hk_step_sd_kernel (1024 workgroups x 256 threads,
// grid-stride over trajectories, 128-VGPR budget), no elimination.
//   A  row-major D x D, thread (ti,tj) owns (16 ra + ti, 16 rb + tj): 8 B per lane, 128-B row segments, pitch 8 D
//   B  the same with two adjacent columns per lane: 16 B per lane, 256-B row segments
//   C  linear: the 4 D^2 doubles of a trajectory as one contiguous array, 16 B per lane, fully coalesced
//   D  as A with the row pitch padded to 64 doubles (aligned 128-B segments, 6.7 % more bytes at D = 60)
// build: hipcc --offload-arch=gfx950 -O3 -o stream_patterns stream_patterns.hip ; run: ./stream_patterns [n] [D]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NR, bool CONSEC = false>
__global__ __launch_bounds__(256, 4) void pat_a(double *mono, long n, int D, int pitch) {
    // CONSEC: wave w owns the four consecutive rows 4w .. 4w+3 of a slot instead of w, w+4, w+8, w+12
    const int tid = threadIdx.x, tj = tid & 15;
    const int ti = CONSEC ? (tid >> 6) * 4 + ((tid >> 4) & 3) : ((tid >> 4) & 3) * 4 + (tid >> 6);
    const long blk = (long)D * pitch;
    for (long tr = blockIdx.x; tr < n; tr += gridDim.x) {
        double *M = mono + tr * 4 * blk;
#pragma unroll
        for (int ra = 0; ra < NR; ++ra) {
            const int a = 16 * ra + ti;
            double v[4][NR];
#pragma unroll
            for (int rb = 0; rb < NR; ++rb) {
                const bool ok = a < D && 16 * rb + tj < D;
                const long e = (long)a * pitch + 16 * rb + tj;
#pragma unroll
                for (int p = 0; p < 4; ++p) v[p][rb] = ok ? M[p * blk + e] : 0.0;
            }
#pragma unroll
            for (int rb = 0; rb < NR; ++rb) {
                const bool ok = a < D && 16 * rb + tj < D;
                const long e = (long)a * pitch + 16 * rb + tj;
                const double q = v[0][rb] * 1.0000001 + v[2][rb] * 1e-9, r = v[2][rb] * 1.0000001 - v[0][rb] * 1e-9;
                const double s = v[1][rb] * 1.0000001 + v[3][rb] * 1e-9, t = v[3][rb] * 1.0000001 - v[1][rb] * 1e-9;
                if (ok) { M[e] = q; M[blk + e] = s; M[2 * blk + e] = r; M[3 * blk + e] = t; }
            }
        }
    }
}

// two adjacent columns per lane: thread (ti,tj) owns columns 32 rb + 2 tj, +1
template <int NR, int NC>
__global__ __launch_bounds__(256, 4) void pat_b(double *mono, long n, int D) {
    const int tid = threadIdx.x, ti = ((tid >> 4) & 3) * 4 + (tid >> 6), tj = tid & 15;
    const long blk = (long)D * D;
    for (long tr = blockIdx.x; tr < n; tr += gridDim.x) {
        double *M = mono + tr * 4 * blk;
#pragma unroll
        for (int ra = 0; ra < NR; ++ra) {
            const int a = 16 * ra + ti;
            double2 v[4][NC];
#pragma unroll
            for (int rb = 0; rb < NC; ++rb) {
                const int c = 32 * rb + 2 * tj;
                const bool ok = a < D && c < D;
                const long e = (long)a * D + c;
#pragma unroll
                for (int p = 0; p < 4; ++p) v[p][rb] = ok ? *(const double2 *)(M + p * blk + e) : make_double2(0, 0);
            }
#pragma unroll
            for (int rb = 0; rb < NC; ++rb) {
                const int c = 32 * rb + 2 * tj;
                const bool ok = a < D && c < D;
                const long e = (long)a * D + c;
                double2 o[4];
                o[0].x = v[0][rb].x * 1.0000001 + v[2][rb].x * 1e-9; o[0].y = v[0][rb].y * 1.0000001 + v[2][rb].y * 1e-9;
                o[2].x = v[2][rb].x * 1.0000001 - v[0][rb].x * 1e-9; o[2].y = v[2][rb].y * 1.0000001 - v[0][rb].y * 1e-9;
                o[1].x = v[1][rb].x * 1.0000001 + v[3][rb].x * 1e-9; o[1].y = v[1][rb].y * 1.0000001 + v[3][rb].y * 1e-9;
                o[3].x = v[3][rb].x * 1.0000001 - v[1][rb].x * 1e-9; o[3].y = v[3][rb].y * 1.0000001 - v[1][rb].y * 1e-9;
                if (ok) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) *(double2 *)(M + p * blk + e) = o[p];
                }
            }
        }
    }
}

// linear: thread t handles double2 chunks t, t + 256, ... of each block; the four blocks of a chunk together
__global__ __launch_bounds__(256, 4) void pat_c(double *mono, long n, int D) {
    const long blk = (long)D * D, nch = blk / 2;
    for (long tr = blockIdx.x; tr < n; tr += gridDim.x) {
        double *M = mono + tr * 4 * blk;
        for (long c0 = 0; c0 < nch; c0 += 256 * 4) {
            double2 v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long c = c0 + u * 256 + threadIdx.x;
#pragma unroll
                for (int p = 0; p < 4; ++p) v[p][u] = c < nch ? *(const double2 *)(M + p * blk + 2 * c) : make_double2(0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long c = c0 + u * 256 + threadIdx.x;
                double2 o[4];
                o[0].x = v[0][u].x * 1.0000001 + v[2][u].x * 1e-9; o[0].y = v[0][u].y * 1.0000001 + v[2][u].y * 1e-9;
                o[2].x = v[2][u].x * 1.0000001 - v[0][u].x * 1e-9; o[2].y = v[2][u].y * 1.0000001 - v[0][u].y * 1e-9;
                o[1].x = v[1][u].x * 1.0000001 + v[3][u].x * 1e-9; o[1].y = v[1][u].y * 1.0000001 + v[3][u].y * 1e-9;
                o[3].x = v[3][u].x * 1.0000001 - v[1][u].x * 1e-9; o[3].y = v[3][u].y * 1.0000001 - v[1][u].y * 1e-9;
                if (c < nch) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) *(double2 *)(M + p * blk + 2 * c) = o[p];
                }
            }
        }
    }
}

// linear, 8 B per lane: what a wave-tiled storage order of the blocks would give the 16 x 16 thread grid
// (every wave instruction reads 512 contiguous bytes, the four waves 2 KB, the workgroup walks the block linearly)
__global__ __launch_bounds__(256, 4) void pat_e(double *mono, long n, int D) {
    const long blk = (long)D * D;
    for (long tr = blockIdx.x; tr < n; tr += gridDim.x) {
        double *M = mono + tr * 4 * blk;
        for (long c0 = 0; c0 < blk; c0 += 256 * 4) {
            double v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long c = c0 + u * 256 + threadIdx.x;
#pragma unroll
                for (int p = 0; p < 4; ++p) v[p][u] = c < blk ? M[p * blk + c] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long c = c0 + u * 256 + threadIdx.x;
                const double q = v[0][u] * 1.0000001 + v[2][u] * 1e-9, r = v[2][u] * 1.0000001 - v[0][u] * 1e-9;
                const double s = v[1][u] * 1.0000001 + v[3][u] * 1e-9, t = v[3][u] * 1.0000001 - v[1][u] * 1e-9;
                if (c < blk) { M[c] = q; M[blk + c] = s; M[2 * blk + c] = r; M[3 * blk + c] = t; }
            }
        }
    }
}

// pattern E with non-temporal loads and stores
__global__ __launch_bounds__(256, 4) void pat_e_nt(double *mono, long n, int D) {
    const long blk = (long)D * D;
    for (long tr = blockIdx.x; tr < n; tr += gridDim.x) {
        double *M = mono + tr * 4 * blk;
        for (long c0 = 0; c0 < blk; c0 += 256 * 4) {
            double v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long c = c0 + u * 256 + threadIdx.x;
#pragma unroll
                for (int p = 0; p < 4; ++p) v[p][u] = c < blk ? __builtin_nontemporal_load(&M[p * blk + c]) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long c = c0 + u * 256 + threadIdx.x;
                const double q = v[0][u] * 1.0000001 + v[2][u] * 1e-9, r = v[2][u] * 1.0000001 - v[0][u] * 1e-9;
                const double s = v[1][u] * 1.0000001 + v[3][u] * 1e-9, t = v[3][u] * 1.0000001 - v[1][u] * 1e-9;
                if (c < blk) {
                    __builtin_nontemporal_store(q, &M[c]); __builtin_nontemporal_store(s, &M[blk + c]);
                    __builtin_nontemporal_store(r, &M[2 * blk + c]); __builtin_nontemporal_store(t, &M[3 * blk + c]);
                }
            }
        }
    }
}


// ---- round 3: is the in-place pattern the floor?  The same bytes read from one buffer and written to ANOTHER ----
// pattern E out of place: reads src, writes dst (same thread -> address mapping, same launch shape)
template <bool NT>
__global__ __launch_bounds__(256, 4) void pat_e_oop(const double *src, double *dst, long n, int D) {
    const long blk = (long)D * D;
    for (long tr = blockIdx.x; tr < n; tr += gridDim.x) {
        const double *M = src + tr * 4 * blk;
        double *O = dst + tr * 4 * blk;
        for (long c0 = 0; c0 < blk; c0 += 256 * 4) {
            double v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long c = c0 + u * 256 + threadIdx.x;
#pragma unroll
                for (int p = 0; p < 4; ++p) v[p][u] = c < blk ? (NT ? __builtin_nontemporal_load(&M[p * blk + c]) : M[p * blk + c]) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long c = c0 + u * 256 + threadIdx.x;
                const double q = v[0][u] * 1.0000001 + v[2][u] * 1e-9, r = v[2][u] * 1.0000001 - v[0][u] * 1e-9;
                const double s = v[1][u] * 1.0000001 + v[3][u] * 1e-9, t = v[3][u] * 1.0000001 - v[1][u] * 1e-9;
                if (c < blk) {
                    if (NT) {
                        __builtin_nontemporal_store(q, &O[c]); __builtin_nontemporal_store(s, &O[blk + c]);
                        __builtin_nontemporal_store(r, &O[2 * blk + c]); __builtin_nontemporal_store(t, &O[3 * blk + c]);
                    } else { O[c] = q; O[blk + c] = s; O[2 * blk + c] = r; O[3 * blk + c] = t; }
                }
            }
        }
    }
}

// plain copy, 16 B per lane, grid-stride over the whole array (the guide's "float4 copy" figure on THIS box)
typedef double sc_d2 __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ __launch_bounds__(256, 4) void copy16(const sc_d2 *src, sc_d2 *dst, long count) {
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i0 = (long)blockIdx.x * 256 * 4; i0 < count; i0 += stride) {
        sc_d2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = i0 + u * 256 + threadIdx.x;
            v[u] = i < count ? (NT ? __builtin_nontemporal_load(&src[i]) : src[i]) : sc_d2{0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = i0 + u * 256 + threadIdx.x;
            if (i < count) { if (NT) __builtin_nontemporal_store(v[u], &dst[i]); else dst[i] = v[u]; }
        }
    }
}

// read only (sum to one value per thread, written once) and write only: the two halves of the stream on their own
__global__ __launch_bounds__(256, 4) void read16(const double2 *src, double *out, long count) {
    const long stride = (long)gridDim.x * 256 * 4;
    double acc = 0.0;
    for (long i0 = (long)blockIdx.x * 256 * 4; i0 < count; i0 += stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = i0 + u * 256 + threadIdx.x;
            if (i < count) { const double2 v = src[i]; acc += v.x + v.y; }
        }
    }
    if (acc == 12345.678) out[0] = acc;
}
__global__ __launch_bounds__(256, 4) void write16(double2 *dst, long count) {
    const long stride = (long)gridDim.x * 256 * 4;
    for (long i0 = (long)blockIdx.x * 256 * 4; i0 < count; i0 += stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = i0 + u * 256 + threadIdx.x;
            if (i < count) dst[i] = make_double2(1.0, 2.0);
        }
    }
}

template <class F>
static float timed(F f, int reps = 5) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 100000;
    const int D = argc > 2 ? atoi(argv[2]) : 60;
    const int grid = 1024;
    double *m;
    const size_t bytes = (size_t)n * 4 * 64 * 64 * 8;      // room for the padded variant
    CHECK(hipMalloc(&m, bytes));
    CHECK(hipMemset(m, 0, bytes));
    const double gb = (double)n * 4 * D * D * 8 * 2 / 1e9;
    float t;
    t = timed([&] { hipLaunchKernelGGL(pat_a<4>, dim3(grid), dim3(256), 0, 0, m, n, D, D); });
    printf("A  8 B/lane, pitch %d      : %.3f ms  %.0f GB/s\n", D, t, gb / t * 1e3);
    t = timed([&] { hipLaunchKernelGGL((pat_b<4, 2>), dim3(grid), dim3(256), 0, 0, m, n, D); });
    printf("B 16 B/lane, two columns   : %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
    t = timed([&] { hipLaunchKernelGGL(pat_c, dim3(grid), dim3(256), 0, 0, m, n, D); });
    printf("C 16 B/lane, linear        : %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
    t = timed([&] { hipLaunchKernelGGL(pat_a<4>, dim3(grid), dim3(256), 0, 0, m, n, D, 64); });
    printf("D  8 B/lane, pitch 64      : %.3f ms  %.0f GB/s (useful bytes; %.0f GB/s moved)\n", t, gb / t * 1e3, gb / t * 1e3 * 64 / D);
    t = timed([&] { hipLaunchKernelGGL((pat_a<4, true>), dim3(grid), dim3(256), 0, 0, m, n, D, D); });
    printf("A' consecutive rows / wave : %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
    t = timed([&] { hipLaunchKernelGGL(pat_e, dim3(grid), dim3(256), 0, 0, m, n, D); });
    printf("E  8 B/lane, linear        : %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
    for (int g : {512, 2048, 4096}) {
        t = timed([&] { hipLaunchKernelGGL(pat_e, dim3(g), dim3(256), 0, 0, m, n, D); });
        printf("E grid %4d                : %.3f ms  %.0f GB/s\n", g, t, gb / t * 1e3);
        t = timed([&] { hipLaunchKernelGGL(pat_c, dim3(g), dim3(256), 0, 0, m, n, D); });
        printf("C grid %4d                : %.3f ms  %.0f GB/s\n", g, t, gb / t * 1e3);
        t = timed([&] { hipLaunchKernelGGL(pat_a<4>, dim3(g), dim3(256), 0, 0, m, n, D, D); });
        printf("A grid %4d                : %.3f ms  %.0f GB/s\n", g, t, gb / t * 1e3);
    }
    for (int g : {1024, 4096, 8192}) {
        t = timed([&] { hipLaunchKernelGGL(pat_e_nt, dim3(g), dim3(256), 0, 0, m, n, D); });
        printf("E non-temporal, grid %4d  : %.3f ms  %.0f GB/s\n", g, t, gb / t * 1e3);
    }

    // ---- round 3: out of place, same session ----
    {
        double *m2;
        const size_t used = (size_t)n * 4 * D * D * 8;
        CHECK(hipMalloc(&m2, used));
        CHECK(hipMemset(m2, 0, used));
        const long cnt = (long)(used / 16);
        for (int g : {1024, 2048, 4096, 8192}) {
            t = timed([&] { hipLaunchKernelGGL(pat_e, dim3(g), dim3(256), 0, 0, m, n, D); });
            printf("E in place,      grid %4d : %.3f ms  %.0f GB/s\n", g, t, gb / t * 1e3);
            t = timed([&] { hipLaunchKernelGGL(pat_e_oop<false>, dim3(g), dim3(256), 0, 0, m, m2, n, D); });
            printf("E OUT of place,  grid %4d : %.3f ms  %.0f GB/s\n", g, t, gb / t * 1e3);
            t = timed([&] { hipLaunchKernelGGL(pat_e_oop<true>, dim3(g), dim3(256), 0, 0, m, m2, n, D); });
            printf("E OUT of place nt, grid %4d: %.3f ms  %.0f GB/s\n", g, t, gb / t * 1e3);
            t = timed([&] { hipLaunchKernelGGL(copy16<false>, dim3(g), dim3(256), 0, 0, (const sc_d2 *)m, (sc_d2 *)m2, cnt); });
            printf("plain 16-B copy, grid %4d : %.3f ms  %.0f GB/s\n", g, t, gb / t * 1e3);
            t = timed([&] { hipLaunchKernelGGL(copy16<true>, dim3(g), dim3(256), 0, 0, (const sc_d2 *)m, (sc_d2 *)m2, cnt); });
            printf("plain 16-B copy nt, grid %4d: %.3f ms  %.0f GB/s\n", g, t, gb / t * 1e3);
        }
        t = timed([&] { hipLaunchKernelGGL(copy16<false>, dim3(65536), dim3(256), 0, 0, (const sc_d2 *)m, (sc_d2 *)m2, cnt); });
        printf("plain 16-B copy, grid 65536: %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
        t = timed([&] { CHECK(hipMemcpyAsync(m2, m, used, hipMemcpyDeviceToDevice, 0)); });
        printf("hipMemcpy D2D              : %.3f ms  %.0f GB/s\n", t, gb / t * 1e3);
        t = timed([&] { hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const double2 *)m, m2, cnt); });
        printf("read only  (11.52 GB)      : %.3f ms  %.0f GB/s\n", t, gb / 2 / t * 1e3);
        t = timed([&] { hipLaunchKernelGGL(write16, dim3(4096), dim3(256), 0, 0, (double2 *)m2, cnt); });
        printf("write only (11.52 GB)      : %.3f ms  %.0f GB/s\n", t, gb / 2 / t * 1e3);
        CHECK(hipFree(m2));
    }
    CHECK(hipMemcpy(m, m + 1, 8, hipMemcpyDeviceToDevice));
    hipFree(m);
    return 0;
}
