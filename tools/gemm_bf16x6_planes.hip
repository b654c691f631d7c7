// Prototype 2 of the fp32-on-bf16-matrix-cores GEMM (see tools/gemm_bf16x6.hip for the scheme and its
// accuracy): the operands are split ONCE into three bf16 planes by a streaming pass
// (x = hi + mid + lo exactly, truncation cuts) and the GEMM (NT: C[M][N] = A[M][K] . B[N][K]^T) is a
// pure bf16 kernel over the planes -- no VALU work in the k-loop (prototype 1 spent ~130 non-MFMA
// instructions per 24 MFMAs on the in-kernel split and was bound by them, not by the matrix pipe).
//   per k-step (16 k) and thread: 6 x 16-byte global loads (one per operand plane), 6 x ds_write_b128,
//   12 x ds_read_b128, 24 x v_mfma_f32_32x32x16_bf16 per wave (128 x 128 tile, 2 x 2 waves of 64 x 64).
//   LDS image per plane: [k half][row][8 k] (16-byte chunks = fragment elements), second half offset
//   by 128 B modulo the bank row.
//   Global plane layout (TILED=1): [k / 16][k half][row][8 k] -- the same chunk order as the LDS image, so
//   a wave's staging load is 1 KB contiguous.  With row-major planes (TILED=0, first measurement) a wave
//   load touched 32 rows x 32 B: 3x the L1 line requests of prototype 1 and the GEMM was SLOWER than it
//   (logits 938 vs 610 us, square 1535 vs 852 us) although it has no VALU work.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/gemm_bf16x6_planes tools/gemm_bf16x6_planes.hip
// In the product the W planes would be produced once per optimizer step (51 MB -> 77 MB, ~30 us) and
// shared by the logits product and (transposed-read) the head's dX; X planes are 2 MB.
// Measured on MI355X at the end of round 1 (errors identical to prototype 1, i.e. fp32-grade):
//   TILED=1: logits 2752 x 100001 x 128  674 us (split of W + X 26 us), square 4096  808 us = 170 TF/s
//            fp32-equivalent (split 218 us: its transposing writes are uncoalesced, needs an LDS transpose),
//            ff1 45 us.   TILED=0: 938 / 1535 / 69 us.
//   Both prototypes stop at 610-670 us on the K = 128 logits shape: a workgroup has only 8 k-steps of
//   0.3 us each, far less than one global-load latency, and 3 workgroups per CU cannot cover it.  The
//   shape wants a W-tile-stationary persistent workgroup (W tile split / loaded once, X tiles streamed
//   with global_load_lds into a double buffer, stores of tile t under the MFMAs of tile t+1): 67 k
//   64 x 64 x 128 tiles x 0.64 us of MFMA per CU-slot = ~170 us of matrix work for the whole product.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // first-class vector: a ring of HIP uint4 structs stayed in scratch

#define BM 128
#define BN 128
#define BK 16
#ifndef TILED
#define TILED 1
#endif
#define KH_WORDS 544          // 128 rows x 4 words + 32 words (128 B) of stagger
#define PL_WORDS 1088

// ------------------------------------------------------------------------------------------ split pass
// planes p0 | p1 | p2, each [rows][K] bf16 (K % 8 == 0), 8 elements per thread
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, unsigned short* __restrict__ p0,
                                                           unsigned short* __restrict__ p1, unsigned short* __restrict__ p2,
                                                           size_t n8, int rows, int K) {
    const int cpr = K / 8;      // chunks per row
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        // chunk (row, ck) -> TILED: ((ck / 2) * 2 + ck % 2) * rows + row  (= ck * rows + row)
        const size_t o = TILED ? (size_t)(i % cpr) * rows + (size_t)(i / cpr) : i;
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

struct Planes { const unsigned short* p[3]; };

// ------------------------------------------------------------------------------------------ GEMM on planes
__global__ __launch_bounds__(256) void gemm_nt_planes(Planes A, Planes B, float* __restrict__ C, int M, int N, int K,
                                                      int ldc) {
    __shared__ __attribute__((aligned(16))) unsigned S[2][2][3][PL_WORDS];      // [stage][operand][plane]: 51 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, kh = lane >> 5, c = lane & 31;
    const int TM = (M + BM - 1) / BM, TN = (N + BN - 1) / BN;
    int mt, nt;
    {   // XCD-aware order along N
        const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int qd = TN >> 3, rd = TN & 7;
        const int cnt = qd + (x < rd ? 1 : 0), start = x * qd + (x < rd ? x : rd);
        const int il = slot / TM;
        if (il >= cnt) return;
        nt = start + il; mt = slot % TM;
    }
    const int m0 = mt * BM, n0 = nt * BN;
    // staging: thread -> (row = tid / 2, k half = tid % 2): one 16-byte chunk (8 bf16) per plane and operand
#if TILED
    const int lr = tid & 127, lh = tid >> 7;          // a wave = 64 consecutive rows of one k half: 1 KB contiguous
    const long aoff = ((long)lh * M + min(m0 + lr, M - 1)) * 8;      // + kt * 2 * M * 8 per k-step
    const long boff = ((long)lh * N + min(n0 + lr, N - 1)) * 8;
    const long astep = 2L * M * 8, bstep = 2L * N * 8;
#else
    const int lr = tid >> 1, lh = tid & 1;
    const long aoff = (long)min(m0 + lr, M - 1) * K + lh * 8;
    const long boff = (long)min(n0 + lr, N - 1) * K + lh * 8;
    const long astep = BK, bstep = BK;
#endif
    const int sw = lh * KH_WORDS + lr * 4;            // LDS word index of the chunk inside a plane image
    u32x4 ra[3][3], rb[3][3];                          // register ring, 3 k-steps deep x 3 planes
    // (plane pointers as scalars: indexing the by-value argument structs from the lambdas put them, and
    //  with them the whole ring, in scratch)
    const unsigned short* const a0 = A.p[0] + aoff; const unsigned short* const a1 = A.p[1] + aoff;
    const unsigned short* const a2 = A.p[2] + aoff; const unsigned short* const b0 = B.p[0] + boff;
    const unsigned short* const b1 = B.p[1] + boff; const unsigned short* const b2 = B.p[2] + boff;
    auto load = [&](u32x4 (&xa)[3], u32x4 (&xb)[3], int kt) __attribute__((always_inline)) {
        const long ka = (long)kt * astep, kb = (long)kt * bstep;
        xa[0] = *reinterpret_cast<const u32x4*>(a0 + ka); xb[0] = *reinterpret_cast<const u32x4*>(b0 + kb);
        xa[1] = *reinterpret_cast<const u32x4*>(a1 + ka); xb[1] = *reinterpret_cast<const u32x4*>(b1 + kb);
        xa[2] = *reinterpret_cast<const u32x4*>(a2 + ka); xb[2] = *reinterpret_cast<const u32x4*>(b2 + kb);
    };
    auto store = [&](const u32x4 (&xa)[3], const u32x4 (&xb)[3], int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            *reinterpret_cast<u32x4*>(&S[buf][0][pl][sw]) = xa[pl];
            *reinterpret_cast<u32x4*>(&S[buf][1][pl][sw]) = xb[pl];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto multiply = [&](int buf) __attribute__((always_inline)) {
        bf16x8 fa[2][3], fb[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i][pl] = *reinterpret_cast<const bf16x8*>(&S[buf][0][pl][kh * KH_WORDS + (wm * 64 + i * 32 + c) * 4]);
                fb[i][pl] = *reinterpret_cast<const bf16x8*>(&S[buf][1][pl][kh * KH_WORDS + (wn * 64 + i * 32 + c) * 4]);
            }
        // smallest terms first; the four accumulators interleaved (independent chains back to back)
#define TERM(PA, PB)                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)          \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA], fb[j][PB], acc[i][j], 0, 0, 0);
        TERM(1, 1) TERM(2, 0) TERM(0, 2) TERM(1, 0) TERM(0, 1) TERM(0, 0)
#undef TERM
    };
    const int KT = K / BK, last = KT - 1;
    load(ra[0], rb[0], 0);
    load(ra[1], rb[1], min(1, last));
    store(ra[0], rb[0], 0);
    __syncthreads();
    // step kt: request tile kt + 2, park tile kt + 1 in the other LDS buffer, multiply tile kt.
    // The ring roles repeat every 3 steps and the LDS buffers every 2, so the loop body is 6 steps; the
    // last body may run past KT: those steps only re-park a clamped copy of the last tile (loads hit
    // L1) and skip the multiply under a workgroup-uniform branch -- no early exit, so the register
    // roles stay static (with exits between the steps the compiler moved the ring to scratch).
#define STEP(KT_, SLOAD, SNEXT, BUF)                              \
    load(ra[SLOAD], rb[SLOAD], min((KT_) + 2, last));            \
    store(ra[SNEXT], rb[SNEXT], (BUF) ^ 1);                      \
    if ((KT_) < KT) multiply(BUF);                               \
    __syncthreads();
    for (int kt = 0; kt < KT; kt += 6) {
        STEP(kt, 2, 1, 0)
        STEP(kt + 1, 0, 2, 1)
        STEP(kt + 2, 1, 0, 0)
        STEP(kt + 3, 2, 1, 1)
        STEP(kt + 4, 0, 2, 0)
        STEP(kt + 5, 1, 0, 1)
    }
#undef STEP
    const bool rows_full = m0 + BM <= M;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + c;
            if (col >= N) continue;
            float* cp = C + (long)(m0 + wm * 64 + i * 32 + 4 * kh) * ldc + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                if (rows_full || m0 + wm * 64 + i * 32 + 4 * kh + dr < M) cp[(long)dr * ldc] = acc[i][j][r];
            }
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
    struct { const char* name; int M, N, K; } shapes[] = {{"logits 2752 x 100001 x 128", 2752, 100001, 128},
                                                          {"square 4096", 4096, 4096, 4096},
                                                          {"ff1 20480 x 512 x 128", 20480, 512, 128}};
    for (auto& sh : shapes) {
        const int M = sh.M, N = sh.N, K = sh.K, ldc = (N + 63) / 64 * 64;
        float *A, *B, *C;
        unsigned short *Ap, *Bp;
        hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * ldc * 4);
        hipMalloc(&Ap, (size_t)3 * M * K * 2); hipMalloc(&Bp, (size_t)3 * N * K * 2);
        hipMemset(C, 0, (size_t)M * ldc * 4);
        hipLaunchKernelGGL(fill_random, dim3(2048), dim3(256), 0, 0, A, (size_t)M * K, 1u, 1.0f);
        hipLaunchKernelGGL(fill_random, dim3(2048), dim3(256), 0, 0, B, (size_t)N * K, 2u, 0.05f);
        Planes PA{{Ap, Ap + (size_t)M * K, Ap + (size_t)2 * M * K}}, PB{{Bp, Bp + (size_t)N * K, Bp + (size_t)2 * N * K}};
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int reps = 10;
        float ms_split, ms;
        auto split = [&](const float* x, unsigned short* p, int rows) {
            const size_t n = (size_t)rows * K;
            hipLaunchKernelGGL(split_planes_kernel, dim3(4096), dim3(256), 0, 0, x, p, p + n, p + 2 * n, n / 8, rows, K);
        };
        split(A, Ap, M); split(B, Bp, N);
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) { split(A, Ap, M); split(B, Bp, N); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms_split, e0, e1); ms_split /= reps;
        const int TM = (M + BM - 1) / BM, TN = (N + BN - 1) / BN;
        dim3 grid(8 * ((TN + 7) / 8) * TM), block(256);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_nt_planes, grid, block, 0, 0, PA, PB, C, M, N, K, ldc);
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_nt_planes, grid, block, 0, 0, PA, PB, C, M, N, K, ldc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        printf("%-30s GEMM %8.1f us  %6.1f TF/s (fp32-equivalent) | split of both operands %7.1f us | %s\n", sh.name,
               ms * 1e3, 2.0 * M * N * K / ms / 1e9, ms_split * 1e3, hipGetErrorString(hipGetLastError()));
        // accuracy on 8 sampled rows
        const int nr = 8; int hrows[nr]; for (int i = 0; i < nr; ++i) hrows[i] = (int)(((long)i * 7919 + 13) % M);
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
        printf("    error vs fp64 over %d rows (mean |c| %.3e): planes max %.3e mean %.3e | fp32 fma chain max %.3e mean %.3e\n",
               nr, mag / cnt, e_split_max, e_split_sum / cnt, e_f32_max, e_f32_sum / cnt);
        free(h64); free(h32); free(hc);
        hipFree(A); hipFree(B); hipFree(C); hipFree(Ap); hipFree(Bp); hipFree(C64); hipFree(C32); hipFree(drows);
    }
    return 0;
}
