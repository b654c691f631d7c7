// v_mfma_f32_32x32x16_bf16 issue-rate calibration for gfx950 (what tools/mfma_peak.hip does for the fp32 op):
//   pure : CHAINS independent accumulator chains per wave, operands constant or random, no memory traffic
//   lds  : the inner loop of csrc/head_split.hip -- three ds_read_b128 then six dependent MFMAs per K = 16 step,
//          B fragments in registers -- with 1 or 2 accumulators
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_bf16.hip -o /tmp/mfma_peak_bf16 && /tmp/mfma_peak_bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t hash32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
// a random bf16 pair in (-2, 2): exponent 0x3f / 0x3e.., random mantissa
__device__ __forceinline__ uint32_t rnd_bf16x2(uint32_t s) {
    const uint32_t h = hash32(s);
    return (h & 0x807f807fu) | 0x3f003f00u;
}
__device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

template <int CHAINS>
__global__ __launch_bounds__(256) void pure_loop(float* out, int iters, int random) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    uint4 a[4], b[4];
    const uint32_t t = (blockIdx.x * 256 + threadIdx.x) * 64;
    for (int u = 0; u < 4; ++u) {
        a[u] = random ? make_uint4(rnd_bf16x2(t + 8 * u), rnd_bf16x2(t + 8 * u + 1), rnd_bf16x2(t + 8 * u + 2), rnd_bf16x2(t + 8 * u + 3)) : make_uint4(0, 0, 0, 0);
        b[u] = random ? make_uint4(rnd_bf16x2(t + 8 * u + 4), rnd_bf16x2(t + 8 * u + 5), rnd_bf16x2(t + 8 * u + 6), rnd_bf16x2(t + 8 * u + 7)) : make_uint4(0, 0, 0, 0);
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = mfma(a[(u + c) & 3], b[(u + 3 * c) & 3], acc[c]);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.678f) out[0] = s;
}

// KS steps per "block": per step three 16-byte LDS reads and six MFMAs against register-resident B fragments
template <int NACC>
__global__ __launch_bounds__(256) void lds_loop(float* out, int iters) {
    constexpr int KS = 8, CH = 16;
    __shared__ uint4 lds[3 * CH * 32];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, khalf = lane >> 5;
    for (int i = tid; i < 3 * CH * 32; i += 256) lds[i] = make_uint4(rnd_bf16x2(4 * i), rnd_bf16x2(4 * i + 1), rnd_bf16x2(4 * i + 2), rnd_bf16x2(4 * i + 3));
    uint4 Bf[KS][3];
    for (int s = 0; s < KS; ++s)
        for (int pl = 0; pl < 3; ++pl) {
            const uint32_t t = ((blockIdx.x * 256 + tid) * KS + s) * 3 + pl;
            Bf[s][pl] = make_uint4(rnd_bf16x2(4 * t), rnd_bf16x2(4 * t + 1), rnd_bf16x2(4 * t + 2), rnd_bf16x2(4 * t + 3));
        }
    __syncthreads();
    f32x16 acc[NACC];
    for (int c = 0; c < NACC; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint4 a[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[pl] = lds[(pl * CH + 2 * s + khalf) * 32 + l32];
            f32x16& lo = acc[0];
            f32x16& hi = acc[NACC - 1];
            lo = mfma(a[1], Bf[s][1], lo);
            hi = mfma(a[1], Bf[s][0], hi);
            lo = mfma(a[2], Bf[s][0], lo);
            hi = mfma(a[0], Bf[s][1], hi);
            lo = mfma(a[0], Bf[s][2], lo);
            hi = mfma(a[0], Bf[s][0], hi);
        }
        asm volatile("" ::: "memory");
    }
    float s = 0.f;
    for (int c = 0; c < NACC; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.678f) out[0] = s;
}

int main() {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 8192;
    const double fl = 2.0 * 32 * 32 * 16;
    for (int random : {0, 1})
        for (int chains : {1, 2, 4})
            for (int wps : {1, 2, 3, 4}) {
                dim3 grid(256 * wps), block(256);
                float ms = 0.f;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (chains == 1) hipLaunchKernelGGL(pure_loop<1>, grid, block, 0, 0, d, iters, random);
                    else if (chains == 2) hipLaunchKernelGGL(pure_loop<2>, grid, block, 0, 0, d, iters, random);
                    else hipLaunchKernelGGL(pure_loop<4>, grid, block, 0, 0, d, iters, random);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                const double n = (double)grid.x * 4 * iters * 8 * chains;
                printf("pure %s chains %d waves/SIMD %d : %8.3f ms  %7.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", random ? "random" : "zero  ",
                       chains, wps, ms, n * fl / ms / 1e9, ms * 1e-3 * 2.4e9 / (n / 1024));
            }
    for (int nacc : {1, 2})
        for (int wps : {1, 2, 3}) {
            dim3 grid(256 * wps), block(256);
            float ms = 0.f;
            const int it = 4096;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (nacc == 1) hipLaunchKernelGGL(lds_loop<1>, grid, block, 0, 0, d, it);
                else hipLaunchKernelGGL(lds_loop<2>, grid, block, 0, 0, d, it);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double n = (double)grid.x * 4 * it * 8 * 6;
            printf("lds-fed 6-term loop, accumulators %d waves/SIMD %d : %8.3f ms  %7.1f TFLOP/s  (%.1f cycles/MFMA/SIMD)\n", nacc, wps, ms,
                   n * fl / ms / 1e9, ms * 1e-3 * 2.4e9 / (n / 1024));
        }
    return 0;
}
