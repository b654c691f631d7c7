// Prototype 3 of the fp32-on-bf16-matrix-cores product, for the K = 128 vocabulary-wide logits shape
// (C[M][N] = X[M][128] . W[N][128]^T, M = label rows ~2.8 k, N = 100 001): W-STATIONARY.
// Prototypes 1 / 2 (tools/gemm_bf16x6*.hip) stop at 610-670 us on this shape because a workgroup has
// only 8 k-steps of 0.3 us, less than one global-load latency.  Here a workgroup owns 128 columns of W
// for the whole product and walks down all of X:
//   * each wave keeps ITS 32 columns of W -- all 128 k, three bf16 planes -- in registers (96 VGPRs):
//     W is read once (77 MB of planes), never staged through LDS;
//   * per step one 32-row tile of X (all k, three planes: 24 KB) is shared through LDS (double buffered,
//     loaded one step ahead, one barrier per step);
//   * per step and wave: 8 k-blocks x 6 partial products = 48 v_mfma_f32_32x32x16_bf16 (0.64 us) against
//     24 ds_read_b128, 6 global loads, 6 ds_write_b128 and 16 row stores: the matrix pipe is the bound
//     (a bf16 MFMA hides ~5 issue slots), the 1.1 GB of logits leave under the MFMAs of the next tile.
//   matrix work for the C2 product: 782 workgroups x 86 steps x 1536 cycles / 256 CUs = 0.40 M cycles = 168 us.
// Operands are the pre-split planes of prototype 2 in the tiled layout [k/8][row][8] (one chunk = 8 k =
// 16 bytes = one fragment element).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/logits_wstat_bf16x6 tools/logits_wstat_bf16x6.hip
// Measured on MI355X (last GPU seconds of round 1; correct on the first run incl. edge tiles, errors as
// prototypes 1 / 2): full product 594 us (fp32 kernel 760-830 us, prototypes 1 / 2 610 / 674 us), 551 us
// for 768 workgroups.  200 VGPRs -> 2 workgroups per CU, and the next X tile is requested only one
// step (0.64 us of MFMAs) ahead: ~3.5 us per step, i.e. still latency- not matrix-bound.  Next: the X
// tiles through global_load_lds into a 3-4 deep LDS ring (no staging registers: 176 VGPRs -> 3 per CU),
// requested 2-3 steps ahead; 8 waves per workgroup sharing the ring.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// RAW_BARRIER=1: the X tile is parked BEFORE the logits are stored and the step ends with
// s_waitcnt lgkmcnt(0) + a raw s_barrier instead of __syncthreads() (a fence: every wave drains its 16 row
// stores, vmcnt(0), before the barrier).  Measured with the round's last GPU seconds: 588 us / 529 us
// (594 / 551 with __syncthreads), same errors -- the store drain is NOT what makes a step 3.5 us.
// Remaining suspects, untested: the single dependent accumulator chain per wave (48 back-to-back
// v_mfma_f32_32x32x16_bf16 into one accumulator, only 2 waves per SIMD to fill its stalls -> try 64-row
// steps = two independent chains sharing the W fragments) and the one-step-ahead tile request.
#ifndef RAW_BARRIER
#define RAW_BARRIER 0
#endif
#define KDIM 128
#define NCK (KDIM / 8)         // 16 chunks of 8 k per row
#define BMS 32                 // rows of X per step
#define BNW 128                // columns of W per workgroup (32 per wave)

// planes p0 | p1 | p2, each [NCK][rows][8] bf16; thread = one chunk (8 consecutive k of one row)
__global__ __launch_bounds__(256) void split_planes_tiled(const float* __restrict__ x, unsigned short* __restrict__ p0,
                                                          unsigned short* __restrict__ p1, unsigned short* __restrict__ p2,
                                                          size_t nchunks, int rows) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nchunks; i += (size_t)gridDim.x * blockDim.x) {
        const size_t o = (size_t)(i % NCK) * rows + (size_t)(i / NCK);
        const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x0 = v[2 * e], x1 = v[2 * e + 1];
            const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
            const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
            const unsigned w0 = __float_as_uint(r0), w1 = __float_as_uint(r1);
            const float s0 = r0 - __uint_as_float(w0 & 0xffff0000u), s1 = r1 - __uint_as_float(w1 & 0xffff0000u);
            h[e] = (u0 >> 16) | (u1 & 0xffff0000u);
            m[e] = (w0 >> 16) | (w1 & 0xffff0000u);
            l[e] = (__float_as_uint(s0) >> 16) | (__float_as_uint(s1) & 0xffff0000u);
        }
        reinterpret_cast<uint4*>(p0)[o] = make_uint4(h[0], h[1], h[2], h[3]);
        reinterpret_cast<uint4*>(p1)[o] = make_uint4(m[0], m[1], m[2], m[3]);
        reinterpret_cast<uint4*>(p2)[o] = make_uint4(l[0], l[1], l[2], l[3]);
    }
}

__global__ __launch_bounds__(256) void logits_wstat(const unsigned short* __restrict__ x0, const unsigned short* __restrict__ x1,
                                                    const unsigned short* __restrict__ x2, const unsigned short* __restrict__ w0,
                                                    const unsigned short* __restrict__ w1, const unsigned short* __restrict__ w2,
                                                    float* __restrict__ C, int M, int N, int ldc, float alpha) {
    // X tile image: [buffer][plane][chunk][32 rows] x 16 B = 2 x 24 KB
    __shared__ __attribute__((aligned(16))) u32x4 S[2][3][NCK][BMS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = lane >> 5, c = lane & 31;
    const int n0 = blockIdx.x * BNW;
    const int col = n0 + wave * 32 + c;
    const int colc = min(col, N - 1);
    // this wave's W fragments: block kt, plane pl -> chunk (2 kt + kh) of row colc
    bf16x8 wf[8][3];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
        const size_t ch = ((size_t)(2 * kt + kh) * N + colc) * 8;
        wf[kt][0] = *reinterpret_cast<const bf16x8*>(w0 + ch);
        wf[kt][1] = *reinterpret_cast<const bf16x8*>(w1 + ch);
        wf[kt][2] = *reinterpret_cast<const bf16x8*>(w2 + ch);
    }
    // staging of the X tile: chunk q = tid + 256 j, j < 6: plane j / 2, chunk index ((j & 1) * 256 + tid) / 32, row tid % 32
    const int srow = tid & 31, sck = tid >> 5;          // sck 0..7 ; second half of the plane: sck + 8
    u32x4 st[6];
    auto load_tile = [&](int step) __attribute__((always_inline)) {
        const size_t r = (size_t)min(step * BMS + srow, M - 1);
        st[0] = *reinterpret_cast<const u32x4*>(x0 + ((size_t)sck * M + r) * 8);
        st[1] = *reinterpret_cast<const u32x4*>(x0 + ((size_t)(sck + 8) * M + r) * 8);
        st[2] = *reinterpret_cast<const u32x4*>(x1 + ((size_t)sck * M + r) * 8);
        st[3] = *reinterpret_cast<const u32x4*>(x1 + ((size_t)(sck + 8) * M + r) * 8);
        st[4] = *reinterpret_cast<const u32x4*>(x2 + ((size_t)sck * M + r) * 8);
        st[5] = *reinterpret_cast<const u32x4*>(x2 + ((size_t)(sck + 8) * M + r) * 8);
    };
    auto park_tile = [&](int buf) __attribute__((always_inline)) {
        S[buf][0][sck][srow] = st[0]; S[buf][0][sck + 8][srow] = st[1];
        S[buf][1][sck][srow] = st[2]; S[buf][1][sck + 8][srow] = st[3];
        S[buf][2][sck][srow] = st[4]; S[buf][2][sck + 8][srow] = st[5];
    };
    const int nsteps = (M + BMS - 1) / BMS;
    load_tile(0);
    park_tile(0);
    __syncthreads();
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        load_tile(min(step + 1, nsteps - 1));                 // in flight under this step's MFMAs
        __builtin_amdgcn_sched_barrier(0);                    // (left alone the scheduler sinks the loads below the MFMAs)
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            const u32x4 a0 = S[buf][0][2 * kt + kh][c], a1 = S[buf][1][2 * kt + kh][c], a2 = S[buf][2][2 * kt + kh][c];
            const bf16x8 f0 = __builtin_bit_cast(bf16x8, a0), f1 = __builtin_bit_cast(bf16x8, a1),
                         f2 = __builtin_bit_cast(bf16x8, a2);
            // smallest terms first
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, wf[kt][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f2, wf[kt][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, wf[kt][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, wf[kt][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, wf[kt][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, wf[kt][0], acc, 0, 0, 0);
        }
#if RAW_BARRIER
        park_tile(buf ^ 1);          // buffer buf ^ 1 was last read in step - 1 (barrier passed)
#endif
        // logits of this 32 x 32 block: rows step*32 + (r & 3) + 8 (r >> 2) + 4 kh, column col
        if (col < N) {
            float* cp = C + (size_t)(step * BMS + 4 * kh) * ldc + col;
            if (step * BMS + BMS <= M) {          // workgroup-uniform: straight-line stores for full tiles
#pragma unroll
                for (int r = 0; r < 16; ++r) cp[(size_t)((r & 3) + 8 * (r >> 2)) * ldc] = alpha * acc[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (step * BMS + 4 * kh + dr < M) cp[(size_t)dr * ldc] = alpha * acc[r];
                }
            }
        }
#if RAW_BARRIER
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#else
        park_tile(buf ^ 1);          // buffer buf ^ 1 was last read in step - 1 (barrier passed)
        __syncthreads();
#endif
    }
}

// ------------------------------------------------------------------------------------------ harness
__global__ void ref_rows(const float* A, const float* B, double* C64, float* C32, const int* rows, int nrows, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, ri = blockIdx.y;
    if (n >= N) return;
    const float* a = A + (long)rows[ri] * K;
    const float* b = B + (long)n * K;
    double s = 0.0; float f = 0.f;
    for (int k = 0; k < K; ++k) { s += (double)a[k] * (double)b[k]; f = fmaf(a[k], b[k], f); }
    C64[(long)ri * N + n] = s; C32[(long)ri * N + n] = f;
}
__global__ void fill_random(float* x, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        x[i] = ((float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f) * scale;
    }
}

int main() {
    struct { const char* name; int M, N; } shapes[] = {{"logits 2752 x 100001 x 128", 2752, 100001},
                                                       {"logits 2752 x 98304 x 128 (768 workgroups = one residency)", 2752, 98304},
                                                       {"small  77 x 1001 x 128 (edge tiles)", 77, 1001}};
    const int K = KDIM;
    for (auto& sh : shapes) {
        const int M = sh.M, N = sh.N, ldc = (N + 63) / 64 * 64;
        float *A, *B, *C;
        unsigned short *Ap, *Bp;
        hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * ldc * 4);
        hipMalloc(&Ap, (size_t)3 * M * K * 2); hipMalloc(&Bp, (size_t)3 * N * K * 2);
        hipMemset(C, 0, (size_t)M * ldc * 4);
        hipLaunchKernelGGL(fill_random, dim3(2048), dim3(256), 0, 0, A, (size_t)M * K, 1u, 1.0f);
        hipLaunchKernelGGL(fill_random, dim3(2048), dim3(256), 0, 0, B, (size_t)N * K, 2u, 0.05f);
        const size_t na = (size_t)M * K, nb = (size_t)N * K;
        hipLaunchKernelGGL(split_planes_tiled, dim3(4096), dim3(256), 0, 0, A, Ap, Ap + na, Ap + 2 * na, na / 8, M);
        hipLaunchKernelGGL(split_planes_tiled, dim3(4096), dim3(256), 0, 0, B, Bp, Bp + nb, Bp + 2 * nb, nb / 8, N);
        dim3 grid((N + BNW - 1) / BNW), block(256);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int reps = 10;
        for (int i = 0; i < 3; ++i)
            hipLaunchKernelGGL(logits_wstat, grid, block, 0, 0, Ap, Ap + na, Ap + 2 * na, Bp, Bp + nb, Bp + 2 * nb, C, M, N, ldc, 1.0f);
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i)
            hipLaunchKernelGGL(logits_wstat, grid, block, 0, 0, Ap, Ap + na, Ap + 2 * na, Bp, Bp + nb, Bp + 2 * nb, C, M, N, ldc, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        printf("%-62s %8.1f us  %6.1f TF/s (fp32-equivalent) | %s\n", sh.name, ms * 1e3, 2.0 * M * N * K / ms / 1e9,
               hipGetErrorString(hipGetLastError()));
        const int nr = 8; int hrows[nr]; for (int i = 0; i < nr; ++i) hrows[i] = (int)(((long)i * 7919 + 13) % M);
        hrows[nr - 1] = M - 1;
        int* drows; hipMalloc(&drows, sizeof(hrows)); hipMemcpy(drows, hrows, sizeof(hrows), hipMemcpyHostToDevice);
        double* C64; float* C32; hipMalloc(&C64, (size_t)nr * N * 8); hipMalloc(&C32, (size_t)nr * N * 4);
        hipLaunchKernelGGL(ref_rows, dim3((N + 255) / 256, nr), dim3(256), 0, 0, A, B, C64, C32, drows, nr, N, K);
        double* h64 = (double*)malloc((size_t)nr * N * 8); float* h32 = (float*)malloc((size_t)nr * N * 4);
        float* hc = (float*)malloc((size_t)N * 4);
        hipMemcpy(h64, C64, (size_t)nr * N * 8, hipMemcpyDeviceToHost); hipMemcpy(h32, C32, (size_t)nr * N * 4, hipMemcpyDeviceToHost);
        double e_split_max = 0, e_f32_max = 0, e_split_sum = 0, e_f32_sum = 0, mag = 0;
        for (int i = 0; i < nr; ++i) {
            hipMemcpy(hc, C + (size_t)hrows[i] * ldc, (size_t)N * 4, hipMemcpyDeviceToHost);
            for (int n = 0; n < N; ++n) {
                const double r = h64[(size_t)i * N + n];
                const double es = fabs((double)hc[n] - r), ef = fabs((double)h32[(size_t)i * N + n] - r);
                e_split_max = fmax(e_split_max, es); e_f32_max = fmax(e_f32_max, ef);
                e_split_sum += es; e_f32_sum += ef; mag += fabs(r);
            }
        }
        const double cnt = (double)nr * N;
        printf("    error vs fp64 over %d rows (mean |c| %.3e): w-stationary max %.3e mean %.3e | fp32 fma chain max %.3e mean %.3e\n",
               nr, mag / cnt, e_split_max, e_split_sum / cnt, e_f32_max, e_f32_sum / cnt);
        free(h64); free(h32); free(hc);
        hipFree(A); hipFree(B); hipFree(C); hipFree(Ap); hipFree(Bp); hipFree(C64); hipFree(C32); hipFree(drows);
    }
    return 0;
}
