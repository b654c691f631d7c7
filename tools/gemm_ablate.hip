// Ablation of the fp32 MFMA GEMM inner structure (NT, 64x64x16 tile, interior tiles only): the same
// three-stage pipeline as transformers4rec_amd/csrc/gemm_f32.hip with parts switched off at compile
// time, to see which part keeps the MFMA pipe at 55-65 % when a pure MFMA loop reaches 99 %.
// Results are meaningless numerically when a part is off; only the time matters.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/gemm_ablate tools/gemm_ablate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { F_LOAD = 1, F_STORE = 2, F_BARRIER = 4, F_READ = 8, F_EPI = 16, F_ALL = 31 };

template <int FLAGS>
__global__ __launch_bounds__(256) void gemm_nt(const float* __restrict__ A, const float* __restrict__ B,
                                               float* __restrict__ C, int M, int N, int K, int xcd) {
    constexpr int BM = 64, BN = 64, BK = 16;
    __shared__ __attribute__((aligned(16))) float As[2][BM * BK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, khalf = lane >> 5;
    const int TM = M / BM, TN = N / BN;
    int mt, nt;
    if (xcd) {   // same XCD-aware order as the product kernel (long dimension = N here)
        const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int qd = TN >> 3, rd = TN & 7;
        const int cnt = qd + (x < rd ? 1 : 0), start = x * qd + (x < rd ? x : rd);
        const int il = slot / TM;
        if (il >= cnt) return;
        nt = start + il; mt = slot % TM;
    } else { mt = blockIdx.x / TN; nt = blockIdx.x % TN; }
    const int m0 = mt * BM, n0 = nt * BN;
    const int lm = tid / 4, lk4 = (tid % 4) * 4;
    auto swz = [](int row, int chunk) { return (chunk ^ ((row >> 2) & 3)) * 4; };
    const float* ap = A + (long)(m0 + lm) * K + lk4;
    const float* bp = B + (long)(n0 + lm) * K + lk4;
    float4 ra0, rb0, ra1, rb1;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int arow = wm * 32 + (lane & 31), bcol = wn * 32 + (lane & 31);
    float4 fa[2], fb[2];
    auto load = [&](float4& ra, float4& rb, int kt) __attribute__((always_inline)) {
        if (FLAGS & F_LOAD) {
            ra = *reinterpret_cast<const float4*>(ap + kt * BK);
            rb = *reinterpret_cast<const float4*>(bp + kt * BK);
        }
    };
    auto store = [&](float4& ra, float4& rb, int buf) __attribute__((always_inline)) {
        asm volatile("" : "+v"(ra.x), "+v"(ra.y), "+v"(ra.z), "+v"(ra.w));
        asm volatile("" : "+v"(rb.x), "+v"(rb.y), "+v"(rb.z), "+v"(rb.w));
        if (FLAGS & F_STORE) {
            *reinterpret_cast<float4*>(&As[buf][lm * BK + swz(lm, lk4 / 4)]) = ra;
            *reinterpret_cast<float4*>(&Bs[buf][lm * BK + swz(lm, lk4 / 4)]) = rb;
        }
    };
    auto readf = [&](int buf, int h) __attribute__((always_inline)) {
        if (FLAGS & F_READ) {
            fa[h] = *reinterpret_cast<const float4*>(&As[buf][arow * BK + swz(arow, khalf * 2 + h)]);
            fb[h] = *reinterpret_cast<const float4*>(&Bs[buf][bcol * BK + swz(bcol, khalf * 2 + h)]);
        }
    };
    auto mfma4 = [&](int h) __attribute__((always_inline)) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h].x, fb[h].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h].y, fb[h].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h].z, fb[h].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h].w, fb[h].w, acc, 0, 0, 0);
    };
    ra0 = rb0 = ra1 = rb1 = make_float4(1.f, 1.f, 1.f, 1.f);
    fa[0] = fa[1] = fb[0] = fb[1] = make_float4(1.f, 2.f, 3.f, 4.f);
    const int KT = K / BK, last = KT - 1;
    load(ra0, rb0, 0);
    load(ra1, rb1, min(1, last));
    __builtin_amdgcn_sched_barrier(0);
    store(ra0, rb0, 0);
    if (FLAGS & F_BARRIER) __syncthreads();
    readf(0, 0); readf(0, 1);
#define STEP(RA_L, RB_L, RA_S, RB_S, BUFN, TN_)                 \
    load(RA_L, RB_L, min((TN_) + 1, last));                     \
    __builtin_amdgcn_sched_barrier(0);                          \
    mfma4(0);                                                   \
    __builtin_amdgcn_sched_barrier(0);                          \
    store(RA_S, RB_S, BUFN);                                    \
    if (FLAGS & F_BARRIER) __syncthreads();                     \
    readf(BUFN, 0);                                             \
    __builtin_amdgcn_sched_barrier(0);                          \
    mfma4(1);                                                   \
    __builtin_amdgcn_sched_barrier(0);                          \
    readf(BUFN, 1);
    for (int kt = 0; kt < KT; kt += 2) {
        STEP(ra0, rb0, ra1, rb1, 1, kt + 1)
        STEP(ra1, rb1, ra0, rb0, 0, kt + 2)
    }
    if (FLAGS & F_EPI) {
        const int col = n0 + wn * 32 + (lane & 31);
        float* c0 = C + (long)(m0 + wm * 32 + 4 * khalf) * N + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) c0[(long)((r & 3) + 8 * (r >> 2)) * N] = acc[r];
    } else {
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += acc[r];
        if (s == 12345.678f) C[0] = s;
    }
}

__global__ void fill_random(float* x, size_t n, unsigned seed) {   // uniform(-1, 1): zero-filled operands run faster (less switching power)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        x[i] = (float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f;
    }
}

template <int FLAGS>
static void run(const char* name, const float* A, const float* B, float* C, int M, int N, int K) {
    const int TM = M / 64, TN = N / 64;
    dim3 grid(8 * ((TN + 7) / 8) * TM), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(gemm_nt<FLAGS>, grid, block, 0, 0, A, B, C, M, N, K, 1);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_nt<FLAGS>, grid, block, 0, 0, A, B, C, M, N, K, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("  %-44s %8.1f us  %6.1f TF/s\n", name, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
}

int main() {
    struct { const char* name; int M, N, K; } shapes[] = {{"square 4096", 4096, 4096, 4096},
                                                          {"logits 2752 x 100032 x 128", 2752, 100032, 128},
                                                          {"ff1 20480 x 512 x 128", 20480, 512, 128}};
    for (auto& sh : shapes) {
        float *A, *B, *C;
        hipMalloc(&A, (size_t)sh.M * sh.K * 4); hipMalloc(&B, (size_t)sh.N * sh.K * 4); hipMalloc(&C, (size_t)sh.M * sh.N * 4);
        hipLaunchKernelGGL(fill_random, dim3(2048), dim3(256), 0, 0, A, (size_t)sh.M * sh.K, 1u);
        hipLaunchKernelGGL(fill_random, dim3(2048), dim3(256), 0, 0, B, (size_t)sh.N * sh.K, 2u);
        printf("%s\n", sh.name);
        run<F_ALL>("everything", A, B, C, sh.M, sh.N, sh.K);
        run<F_ALL & ~F_EPI>("no epilogue stores", A, B, C, sh.M, sh.N, sh.K);
        run<F_ALL & ~F_LOAD>("no global loads", A, B, C, sh.M, sh.N, sh.K);
        run<F_ALL & ~F_BARRIER>("no barriers", A, B, C, sh.M, sh.N, sh.K);
        run<F_ALL & ~(F_STORE | F_LOAD)>("no loads, no LDS stores", A, B, C, sh.M, sh.N, sh.K);
        run<F_ALL & ~(F_STORE | F_LOAD | F_BARRIER)>("no loads, no LDS stores, no barriers", A, B, C, sh.M, sh.N, sh.K);
        run<F_EPI>("MFMAs + epilogue only", A, B, C, sh.M, sh.N, sh.K);
        run<0>("MFMAs only", A, B, C, sh.M, sh.N, sh.K);
        hipFree(A); hipFree(B); hipFree(C);
    }
    return 0;
}
