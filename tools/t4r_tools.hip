// MEASUREMENT INFRASTRUCTURE ONLY -- libt4r_tools.so, NOT part of the product library (transformers4rec_amd/lib/libt4r_hip.so
// holds the hot path; nothing in transformers4rec_amd/ loads this file).  Built by __graft_entry__.build() next to the product
// library; used by bench.py's roofline probes, tools/occupier_curve.py and tests/test_round5_gpu.py.
//
//   t4r_tools_copy     a plain float4 device copy: the byte-moving ceiling of THIS box (MI355X_MICROARCH.md quotes 6.29 TB/s
//                      for this form), the denominator of bench.py's `roofline_gather.frac_of_measured_copy`
//   t4r_tools_occupy   k workgroups that do nothing but HOLD their CUs until a flag is set or a time bound passes: what an
//                      RCCL ring kernel does to the chip (one workgroup per channel, resident for the whole collective),
//                      so that "does the step survive a co-resident all-reduce" can be measured on ONE GPU
//                      (the table all-reduce runs under the body's backward)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 0: plain loads / stores; 1: non-temporal loads and stores (data touched once)
template <int MODE>
__global__ __launch_bounds__(256) void copy_f4_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n4) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    // four independent 16-byte requests in flight per lane
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = MODE ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}

extern "C" int t4r_tools_copy(void* stream, void* dst, const void* src, long n_bytes, int mode, int blocks) {
    if (n_bytes <= 0) return 0;
    if ((n_bytes & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15)) return -1;
    const long n4 = n_bytes / 16;
    const int grid = blocks > 0 ? blocks : 256 * 8;
    if (mode) hipLaunchKernelGGL(copy_f4_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst, n4);
    else hipLaunchKernelGGL(copy_f4_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst, n4);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Every workgroup sleeps in short naps until *flag != 0 or `max_ticks` of the constant 100 MHz counter have passed since it
// started -- the bound is unconditional: this kernel can never outlive max_ticks, whatever happens to the flag.
// lds_bytes of dynamic LDS and 256 / 512 threads shape its footprint on the CU (an RCCL channel: 256 threads, little LDS).
__global__ void occupy_kernel(const int* __restrict__ flag, long long max_ticks, int* __restrict__ seen) {
    extern __shared__ int lds_pad[];
    const long long t0 = (long long)wall_clock64();
    if (threadIdx.x == 0) lds_pad[0] = 0;
    while (true) {
        __builtin_amdgcn_s_sleep(32);
        const int f = __atomic_load_n(flag, __ATOMIC_RELAXED);
        if (f != 0) break;
        if ((long long)wall_clock64() - t0 > max_ticks) break;
    }
    // one visible side effect per workgroup (how many ran, for the caller's assertion)
    if (threadIdx.x == 0 && seen) atomicAdd(seen, 1);
}

// k workgroups of `threads` threads and `lds_bytes` of LDS on `stream`; they end when *flag (device int) becomes non-zero or
// after max_us microseconds (clamped to 50 ms), whichever comes first.  seen (device int, optional) += k when they end.
extern "C" int t4r_tools_occupy(void* stream, int k, int threads, int lds_bytes, const int* flag, long max_us, int* seen) {
    if (k <= 0) return 0;
    if (!flag || threads < 64 || threads > 1024 || (threads & 63) || lds_bytes < 4 || lds_bytes > 160 * 1024) return -1;
    if (max_us < 1) max_us = 1;
    if (max_us > 50000) max_us = 50000;
    (void)hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(occupy_kernel, dim3(k), dim3(threads), (size_t)lds_bytes, (hipStream_t)stream, flag, (long long)max_us * 100, seen);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
