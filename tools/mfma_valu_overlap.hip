// Does VALU work issue in the shadow of matrix instructions?  One wave per SIMD (256-thread workgroups, 1 per CU) or two
// (512 threads); per loop iteration NM independent MFMAs (5 accumulators round robin) and NV independent v_fma_f32,
// interleaved 1 : NV/NM in program order.  Prints cycles per iteration for MFMA only, VALU only, and both.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_valu_overlap tools/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, bool DO_M, bool DO_V, int NV>   // KIND 0: v_mfma_f32_16x16x4_f32, 1: v_mfma_f32_16x16x32_bf16
__global__ __launch_bounds__(512) void k(float* out, int iters, long long* cyc) {
    f32x4 acc[5];
    for (int r = 0; r < 5; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(a + i); hb[i] = (__bf16)(b + i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 20; ++m) {
            if (DO_M) {
                if (KIND == 0) acc[m % 5] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m % 5], 0, 0, 0);
                else acc[m % 5] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[m % 5], 0, 0, 0);
            }
            if (DO_V) {
#pragma unroll
                for (int j = 0; j < NV; ++j) v[(m * NV + j) % 8] = __builtin_fmaf(v[(m * NV + j) % 8], 1.0001f, 0.5f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int r = 0; r < 5; ++r) s += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND, bool M, bool V, int NV>
static double run(int threads, float* out, long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<KIND, M, V, NV>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, M, V, NV>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / iters * 1e3 / 20;      // ns per (MFMA + NV VALU) slot
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    for (int threads : {256, 512}) {
        printf("== %d threads per workgroup (%d wave(s) per SIMD), ns per slot (1 MFMA + NV fma)\n", threads, threads / 256);
        printf("f32 16x16x4  NV=4 : mfma %.1f  valu %.1f  both %.1f\n", run<0, true, false, 4>(threads, out, cyc), run<0, false, true, 4>(threads, out, cyc), run<0, true, true, 4>(threads, out, cyc));
        printf("f32 16x16x4  NV=8 : mfma %.1f  valu %.1f  both %.1f\n", run<0, true, false, 8>(threads, out, cyc), run<0, false, true, 8>(threads, out, cyc), run<0, true, true, 8>(threads, out, cyc));
        printf("bf16 16x16x32 NV=2: mfma %.1f  valu %.1f  both %.1f\n", run<1, true, false, 2>(threads, out, cyc), run<1, false, true, 2>(threads, out, cyc), run<1, true, true, 2>(threads, out, cyc));
        printf("bf16 16x16x32 NV=4: mfma %.1f  valu %.1f  both %.1f\n", run<1, true, false, 4>(threads, out, cyc), run<1, false, true, 4>(threads, out, cyc), run<1, true, true, 4>(threads, out, cyc));
        printf("bf16 16x16x32 NV=8: mfma %.1f  valu %.1f  both %.1f\n", run<1, true, false, 8>(threads, out, cyc), run<1, false, true, 8>(threads, out, cyc), run<1, true, true, 8>(threads, out, cyc));
    }
    return 0;
}
