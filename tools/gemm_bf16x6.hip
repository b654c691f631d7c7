// Prototype: fp32 GEMM (NT: C[M][N] = A[M][K] . B[N][K]^T) on the BF16 matrix cores by exact 3-way
// splitting.  Every fp32 operand is cut into three bf16 pieces by truncation, x = hi + mid + lo EXACTLY
// (each cut removes the 8 leading mantissa bits), and the product keeps the six largest of the nine
// partial products (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid; each exact in fp32, the dropped ones
// are below 2^-24 |a||b|), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs per 16 k
// replace eight fp32 MFMAs per 16 k at 1/16 of the cycles each: 192 vs 512 MFMA cycles per 32x32x16.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/gemm_bf16x6 tools/gemm_bf16x6.hip
// Prints time / effective TFLOP/s on the hot-path shapes and the error against an fp64 reference next
// to the error of a plain fp32 FMA chain (what native fp32 arithmetic gives).
//   LDC_ALIGN=<floats>  row alignment of C (default 4; 64 = 256-byte rows: logits 728 -> 672 us)
//   ABLATE=<bits>       1 no epilogue stores, 2 no global loads, 4 no MFMAs, 8 no splitting, 16 no
//                       fragment reads (compile-time variants 0-8, 15, 23, 31)
// Round-1 results on MI355X: error = that of fp32 arithmetic (mean 2.3e-8 vs 2.7e-8 on the
// logits shape); square 4096 157 TF/s fp32-equivalent, logits 610 us; without MFMAs 417 / 416 us,
// without loads + stores + MFMAs 133 / 234 us: the ~130 non-MFMA instructions per 24 MFMAs bound it
// (interleaving them with sched_group_barrier at 192 VGPRs was slower than this plain order at 160).
// Next: operands pre-split into bf16 planes once per step, global_load_lds staging, W-stationary tiles.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define BM 128
#define BN 128
#define BK 16

// two fp32 -> (packed hi pair, packed mid pair, packed lo pair); truncation cuts, exact
template <int ablate>
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    if (ablate & 8) { hi = __float_as_uint(a); mid = __float_as_uint(b); lo = hi ^ mid; return; }
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    const float ra = a - __uint_as_float(ua & 0xffff0000u), rb = b - __uint_as_float(ub & 0xffff0000u);
    const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
    const float sa = ra - __uint_as_float(va & 0xffff0000u), sb = rb - __uint_as_float(vb & 0xffff0000u);
    hi = __builtin_amdgcn_perm(ub, ua, 0x07060302u);     // {ua[31:16], ub[31:16]} -> low half = a
    mid = __builtin_amdgcn_perm(vb, va, 0x07060302u);
    lo = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
}

template <int ablate>
__global__ __launch_bounds__(256) void gemm_nt_bf16x6(const float* __restrict__ A, const float* __restrict__ B,
                                                      float* __restrict__ C, int M, int N, int K, int ldc) {
    // ablate (timing experiments only): 1 = no epilogue stores, 2 = no global loads, 4 = no MFMAs,
    // 8 = no operand splitting (raw bits parked), 16 = no fragment reads
    // LDS: [stage][operand][plane][k half][row][8 k] bf16: a lane group of a fragment read (one k half,
    // consecutive rows) is 256 contiguous bytes; the second k half starts 128 B later modulo the bank
    // row so that the staging writes of the two halves do not collide.  2 x 2 x 3 x 4352 B = 51 KB
#define PL_WORDS 1088
#define KH_WORDS 544
    __shared__ __attribute__((aligned(16))) unsigned S[2][2][3][PL_WORDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, kh = lane >> 5, c = lane & 31;
    const int TM = (M + BM - 1) / BM, TN = (N + BN - 1) / BN;
    int mt, nt;
    {   // XCD-aware order along N (the long dimension of the logits GEMM)
        const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int qd = TN >> 3, rd = TN & 7;
        const int cnt = qd + (x < rd ? 1 : 0), start = x * qd + (x < rd ? x : rd);
        const int il = slot / TM;
        if (il >= cnt) return;
        nt = start + il; mt = slot % TM;
    }
    const int m0 = mt * BM, n0 = nt * BN;
    // staging: thread -> (row = tid / 4 + 64 * h, k chunk = tid % 4), h = 0, 1 ; one float4 per h and operand
    const int lr = tid >> 2, lc = (tid & 3) * 4;
    const float* ap[2];
    const float* bp[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        ap[h] = A + (long)min(m0 + lr + 64 * h, M - 1) * K + lc;
        bp[h] = B + (long)min(n0 + lr + 64 * h, N - 1) * K + lc;
    }
    // register ring, 4 k-tiles deep: the loads of tile kt + 3 are issued while tile kt is multiplied (the
    // bf16 MFMAs of one tile take ~0.3 us, a global load ~1.5 us)
    float4 ra[4][2], rb[4][2];
    auto load = [&](float4 (&xa)[2], float4 (&xb)[2], int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (ablate & 2) { xa[h] = make_float4(1.f, 2.f, 3.f, 4.f); xb[h] = xa[h]; continue; }
            xa[h] = *reinterpret_cast<const float4*>(ap[h] + kt * BK);
            xb[h] = *reinterpret_cast<const float4*>(bp[h] + kt * BK);
        }
    };
    auto store = [&](const float4 (&xa)[2], const float4 (&xb)[2], int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int w = (tid & 2 ? KH_WORDS : 0) + (lr + 64 * h) * 4 + (tid & 1) * 2;
            unsigned h0, m0_, l0, h1, m1, l1;
            split2<ablate>(xa[h].x, xa[h].y, h0, m0_, l0);
            split2<ablate>(xa[h].z, xa[h].w, h1, m1, l1);
            *reinterpret_cast<uint2*>(&S[buf][0][0][w]) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(&S[buf][0][1][w]) = make_uint2(m0_, m1);
            *reinterpret_cast<uint2*>(&S[buf][0][2][w]) = make_uint2(l0, l1);
            split2<ablate>(xb[h].x, xb[h].y, h0, m0_, l0);
            split2<ablate>(xb[h].z, xb[h].w, h1, m1, l1);
            *reinterpret_cast<uint2*>(&S[buf][1][0][w]) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(&S[buf][1][1][w]) = make_uint2(m0_, m1);
            *reinterpret_cast<uint2*>(&S[buf][1][2][w]) = make_uint2(l0, l1);
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
                if (ablate & 16) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { fa[i][pl][e] = (short)(lane + e); fb[i][pl][e] = (short)(lane - e); }
                    continue;
                }
                fa[i][pl] = *reinterpret_cast<const bf16x8*>(&S[buf][0][pl][kh * KH_WORDS + (wm * 64 + i * 32 + c) * 4]);
                fb[i][pl] = *reinterpret_cast<const bf16x8*>(&S[buf][1][pl][kh * KH_WORDS + (wn * 64 + i * 32 + c) * 4]);
            }
        if (ablate & 4) { acc[0][0][0] += (float)fa[0][0][0] + (float)fb[1][2][1] + (float)fa[1][1][0] + (float)fb[0][1][3]; return; }
        // smallest terms first; the four accumulators interleaved (independent chains back to back)
#define TERM(PA, PB)                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)          \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA], fb[j][PB], acc[i][j], 0, 0, 0);
        TERM(1, 1) TERM(2, 0) TERM(0, 2) TERM(1, 0) TERM(0, 1) TERM(0, 0)
#undef TERM
    };
    const int KT = K / BK, last = KT - 1;      // KT % 4 == 0 in this prototype
    load(ra[0], rb[0], 0);
    load(ra[1], rb[1], min(1, last));
    load(ra[2], rb[2], min(2, last));
    store(ra[0], rb[0], 0);
    __syncthreads();
    // step kt: request tile kt + 3, split + park tile kt + 1 in the other LDS buffer, multiply tile kt.
    // (Variant tried: fragments fetched right after the barrier and the split / LDS stores interleaved
    //  under the MFMAs with sched_group_barrier (1 MFMA : 4 VALU : 1/2 DS write) -- 192 VGPRs, 2 waves per
    //  SIMD instead of 3: logits 678 us, square 875 us, slower than this plain order.)
#define STEP(KT_, SLOAD, SNEXT, BUF)                              \
    load(ra[SLOAD], rb[SLOAD], min((KT_) + 3, last));            \
    store(ra[SNEXT], rb[SNEXT], (BUF) ^ 1);                      \
    multiply(BUF);                                               \
    __syncthreads();
    for (int kt = 0; kt < KT; kt += 4) {
        STEP(kt, 3, 1, 0)
        STEP(kt + 1, 0, 2, 1)
        STEP(kt + 2, 1, 3, 0)
        STEP(kt + 3, 2, 0, 1)
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + c;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < M && (!(ablate & 1) || acc[i][j][r] == 12345.678f)) C[(long)row * ldc + col] = acc[i][j][r];
            }
        }
}

static void launch(int ablate, dim3 grid, dim3 block, const float* A, const float* B, float* C, int M, int N, int K, int ldc) {
    switch (ablate) {
#define CASE(V) case V: hipLaunchKernelGGL(gemm_nt_bf16x6<V>, grid, block, 0, 0, A, B, C, M, N, K, ldc); break;
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(15) CASE(23) CASE(31)
#undef CASE
    }
}

// references for `rows` sampled rows: fp64 and a plain fp32 FMA chain
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
    const int ablate = getenv("ABLATE") ? atoi(getenv("ABLATE")) : 0;
    struct { const char* name; int M, N, K; } shapes[] = {{"logits 2752 x 100001 x 128", 2752, 100001, 128},
                                                          {"square 4096", 4096, 4096, 4096},
                                                          {"ff1 20480 x 512 x 128", 20480, 512, 128}};
    for (auto& sh : shapes) {
        const int M = sh.M, N = sh.N, K = sh.K, ldc = getenv("LDC_ALIGN") ? (N + atoi(getenv("LDC_ALIGN")) - 1) / atoi(getenv("LDC_ALIGN")) * atoi(getenv("LDC_ALIGN")) : ((N + 3) & ~3);
        float *A, *B, *C;
        hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * ldc * 4);
        hipLaunchKernelGGL(fill_random, dim3(2048), dim3(256), 0, 0, A, (size_t)M * K, 1u, 1.0f);
        hipLaunchKernelGGL(fill_random, dim3(2048), dim3(256), 0, 0, B, (size_t)N * K, 2u, 0.05f);
        const int TM = (M + BM - 1) / BM, TN = (N + BN - 1) / BN;
        dim3 grid(8 * ((TN + 7) / 8) * TM), block(256);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) launch(ablate, grid, block, A, B, C, M, N, K, ldc);
        hipEventRecord(e0);
        const int reps = 10;
        for (int i = 0; i < reps; ++i) launch(ablate, grid, block, A, B, C, M, N, K, ldc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        printf("%-30s %8.1f us  %6.1f TF/s (fp32-equivalent)\n", sh.name, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
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
        printf("    error vs fp64 over %d rows (mean |c| %.3e): bf16x6 max %.3e mean %.3e | fp32 fma chain max %.3e mean %.3e\n",
               nr, mag / cnt, e_split_max, e_split_sum / cnt, e_f32_max, e_f32_sum / cnt);
        free(h64); free(h32); free(hc);
        hipFree(A); hipFree(B); hipFree(C); hipFree(C64); hipFree(C32); hipFree(drows);
    }
    return 0;
}
