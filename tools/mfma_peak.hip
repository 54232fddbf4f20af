// What v_mfma_f32_32x32x2_f32 sustains on this GPU, and what it costs to put other work next to it.
//   1. register-only MFMA loop, 1 / 2 / 4 accumulators per wave, 1..4 waves per SIMD, 5 ms and 300 ms (a
//      power-limited clock would show in the long run): 154 TFLOP/s, from ONE wave per SIMD.
//   2. the loop as the igemm kernels issue it -- per MFMA a fragment pair from LDS, waited for one MFMA later --
//      plus FILL independent VALU instructions per MFMA: the fp32 matrix instruction runs on the fp32 FMA lanes,
//      so vector-ALU work is not hidden behind it: 2 VALU instructions per MFMA cost a third of the throughput
//      at one wave per SIMD, ~4.5 cycles each at four.  (LDS reads alone are free: 144-152 TFLOP/s.)
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ACC, bool LDS>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float seed) {
    __shared__ float tile[64 * 33];
    const int lane = threadIdx.x & 63;
    if (LDS) {
        for (int i = threadIdx.x; i < 64 * 33; i += 256) tile[i] = seed * (float)(i & 7);
        __syncthreads();
    }
    f32x16 acc[ACC];
#pragma unroll
    for (int a = 0; a < ACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float a0 = seed * lane, b0 = seed + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (LDS) {
                a0 = tile[(u * 2 + (lane >> 5)) * 33 + (lane & 31)];
                b0 = tile[(32 + u * 2 + (lane >> 5)) * 33 + (lane & 31)];
            }
#pragma unroll
            for (int a = 0; a < ACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[a], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < ACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.678f) out[0] = s;      // keeps the loop alive
}

// The same loop as the igemm kernels issue it: per MFMA one fragment pair from LDS at an address that moves with
// the iteration (so the reads stay inside the loop), waited for one MFMA later, plus FILL independent VALU
// instructions, with the MFMAs cycling over ACC accumulators.
template <int ACC, int FILL>
__global__ __launch_bounds__(256) void mfma_fed(float *out, int iters, float seed) {
    __shared__ float tile[64 * 33 + 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 33 + 64; i += 256) tile[i] = seed * (float)(i & 7);
    __syncthreads();
    f32x16 acc[ACC];
#pragma unroll
    for (int a = 0; a < ACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float fill = seed;
    const float *base = tile + (lane >> 5) * 33 + (lane & 31);
    float a0 = base[0], b0 = base[32 * 33];
    for (int it = 0; it < iters; ++it) {
        const float *bp = base + (it & 15);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float a1 = bp[(u * 2) * 33], b1 = bp[(32 + u * 2) * 33];
            __builtin_amdgcn_sched_barrier(0);
            acc[u % ACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[u % ACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < FILL; ++f) fill = fill * 1.0001f + seed;
            __builtin_amdgcn_sched_barrier(0);
            a0 = a1; b0 = b1;
        }
    }
    float s = fill;
#pragma unroll
    for (int a = 0; a < ACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.678f) out[0] = s;
}

template <int ACC, int FILL>
static double run_fed(int blocks_per_cu, int cus, double target_ms, float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = cus * blocks_per_cu;
    int iters = 2000;
    float ms = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mfma_fed<ACC, FILL>), dim3(grid), dim3(256), 0, 0, out, iters, 1e-30f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (pass == 0) iters = (int)(iters * target_ms / (ms > 1e-3f ? ms : 1e-3f)) + 1;
    }
    const double flops = (double)grid * 4 * iters * 8.0 * 4096.0;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return flops / (ms * 1e-3) / 1e12;
}

template <int ACC, bool LDS>
static double run(int blocks_per_cu, int cus, double target_ms, float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = cus * blocks_per_cu;
    int iters = 2000;
    float ms = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mfma_loop<ACC, LDS>), dim3(grid), dim3(256), 0, 0, out, iters, 1e-30f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (pass == 0) iters = (int)(iters * target_ms / (ms > 1e-3f ? ms : 1e-3f)) + 1;
    }
    const double flops = (double)grid * 4 * iters * 8.0 * ACC * 4096.0;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return flops / (ms * 1e-3) / 1e12;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("device %s  CUs %d  clock %.0f MHz (reported max)\n", p.name, cus, p.clockRate / 1000.0);
    float *out;
    hipMalloc(&out, 4096);
    printf("register-only MFMA loop (TFLOP/s)   %8s %8s %8s %8s\n", "1 w/SIMD", "2", "3", "4");
    printf("1 accumulator, 5 ms                 %8.1f %8.1f %8.1f %8.1f\n", run<1, false>(1, cus, 5, out), run<1, false>(2, cus, 5, out), run<1, false>(3, cus, 5, out), run<1, false>(4, cus, 5, out));
    printf("2 accumulators, 5 ms                %8.1f %8.1f %8.1f %8.1f\n", run<2, false>(1, cus, 5, out), run<2, false>(2, cus, 5, out), run<2, false>(3, cus, 5, out), run<2, false>(4, cus, 5, out));
    printf("4 accumulators, 5 ms                %8.1f %8.1f %8.1f %8.1f\n", run<4, false>(1, cus, 5, out), run<4, false>(2, cus, 5, out), run<4, false>(3, cus, 5, out), run<4, false>(4, cus, 5, out));
    printf("4 accumulators, 300 ms (sustained)  %8.1f %8.1f %8.1f %8.1f\n", run<4, false>(1, cus, 300, out), run<4, false>(2, cus, 300, out), run<4, false>(3, cus, 300, out), run<4, false>(4, cus, 300, out));
    printf("%-44s %8s %8s %8s %8s\n", "LDS-fed loop (TFLOP/s, 20 ms)", "1 w/SIMD", "2", "3", "4");
#define ROW(A, F) printf("acc%d, %d VALU fillers per MFMA %16s %8.1f %8.1f %8.1f %8.1f\n", A, F, "", \
        run_fed<A, F>(1, cus, 20, out), run_fed<A, F>(2, cus, 20, out), run_fed<A, F>(3, cus, 20, out), run_fed<A, F>(4, cus, 20, out));
    ROW(1, 0) ROW(1, 2) ROW(1, 8)
    ROW(2, 0) ROW(2, 2) ROW(2, 8)
    ROW(4, 0) ROW(4, 2) ROW(4, 8) ROW(4, 14)
    hipFree(out);
    return 0;
}
