// MFMA issue-rate calibration for gfx950: v_mfma_f32_32x32x2_f32 with no memory traffic.
//   variant 0: one dependent accumulator chain per wave      (what a 32x32 wave tile does)
//   variant 1: four independent accumulator chains per wave
// waves per SIMD is set by the launch (blocks of 256 threads = 1 wave per SIMD per block).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.678f) out[0] = s;
}

// same loop with uniform(-1, 1) operands, different per lane and per step: the sustained rate with real
// data (switching power), which is what a GEMM on random inputs can reach
__device__ __forceinline__ float hash_unit(unsigned h) {
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f;
}
__global__ __launch_bounds__(256) void mfma_loop_random(float* out, int iters, int zero) {
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a[8], b[8];
    for (int u = 0; u < 8; ++u) {
        a[u] = zero ? 0.f : hash_unit((blockIdx.x * 256 + threadIdx.x) * 16 + u);
        b[u] = zero ? 0.f : hash_unit((blockIdx.x * 256 + threadIdx.x) * 16 + 8 + u);
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + c) & 7], b[(u + 3 * c) & 7], acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.678f) out[0] = s;
}

int main() {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    for (int chains : {1, 4}) {
        for (int wps : {1, 2, 4, 8}) {   // waves per SIMD
            dim3 grid(256 * wps), block(256);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (chains == 1) hipLaunchKernelGGL(mfma_loop<1>, grid, block, 0, 0, d, iters, 1.f, 2.f);
                else hipLaunchKernelGGL(mfma_loop<4>, grid, block, 0, 0, d, iters, 1.f, 2.f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)grid.x * 4 * iters * 8 * chains * 4096.0;
            printf("chains %d waves/SIMD %d : %8.3f ms  %7.1f TFLOP/s\n", chains, wps, ms, flops / ms / 1e9);
        }
    }
    for (int zero : {1, 0}) {
        for (int wps : {1, 4}) {
            dim3 grid(256 * wps), block(256);
            float ms = 0.f;
            for (int rep = 0; rep < 3; ++rep) {     // ~0.1-0.4 s each: long enough for the clocks to settle
                hipEventRecord(e0);
                hipLaunchKernelGGL(mfma_loop_random, grid, block, 0, 0, d, iters * 16, zero);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double flops = (double)grid.x * 4 * iters * 16 * 8 * 4 * 4096.0;
            printf("%s operands, 4 chains, waves/SIMD %d : %8.3f ms  %7.1f TFLOP/s\n", zero ? "zero  " : "random", wps, ms,
                   flops / ms / 1e9);
        }
    }
    return 0;
}
