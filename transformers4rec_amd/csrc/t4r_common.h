// Shared device/host helpers for the t4r_hip kernels (gfx950 / CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#define T4R_WAVE 64

// ---- environment switches.  The PRODUCT library reads ONE environment variable, T4R_GEMM_PREC (the default precision mode;
// INTEGRATION.md section 5 lists it with the five the Python host side reads).  Every other T4R_* name in these sources is an
// A/B, tuning or fallback-forcing switch of an EXPERIMENT build (-DT4R_EXPERIMENTAL: tools/experimental/build_variant.sh):
// in the product build t4r_exp_getenv() returns NULL, so the compiled-in default -- the measured best -- is what runs, and no
// untested combination of kernel families can be selected from outside.  (The kernels behind those switches that stay in
// the library are the general-shape paths the fast ones fall back to -- other d_model / d_head / sequence lengths,
// misaligned operands --, not alternatives for the same shape.)
#include <stdlib.h>
static inline const char* t4r_exp_getenv(const char* name) {
#ifdef T4R_EXPERIMENTAL
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

extern "C" void t4r_set_error(const char* msg);

#define T4R_CHECK_ARG(cond, msg)          \
    do {                                  \
        if (!(cond)) {                    \
            t4r_set_error(msg);           \
            return -1;                    \
        }                                 \
    } while (0)

// The maximum-dynamic-LDS attribute of a kernel is set per DEVICE: a process-wide "set it once" flag leaves the second device of
// a process without it and its launches fail (several devices per process: xlnet_layer.hip keeps per-device side streams).
// One word per device holds the largest size already set for this kernel there; two racing threads at worst set it twice.
#define T4R_MAX_DEVICES 16
struct T4rLdsAttr { std::atomic<unsigned> bytes[T4R_MAX_DEVICES]; };
static inline void t4r_ensure_dynamic_lds(const void* kernel, size_t smem, T4rLdsAttr& a) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<unsigned>& have = a.bytes[dev & (T4R_MAX_DEVICES - 1)];
    if (smem > have.load(std::memory_order_relaxed)) {
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        have.store((unsigned)smem, std::memory_order_relaxed);
    }
}

#define T4R_LAUNCH_CHECK()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) {                            \
            t4r_set_error(hipGetErrorString(e__));          \
            return (int)e__;                                \
        }                                                   \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// erf(x), branch-free, <= 1 ulp (two minimax fits, N. Juffa's single-precision erff: |x| <= 0.927734375 an odd
// polynomial in x, beyond it 1 - exp(p(|x|)) -- both evaluated, one selected: ~20 VALU ops, no divergent branches;
// the ocml erff is a three-way branch per element, ~3x the issue slots inside a GEMM epilogue).
__device__ __forceinline__ float erf_bf(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    r = 1.0f - __expf(r);               // r <= -0.92 here when selected: exp2(r log2 e) is good to ~2e-7 relative
    r = copysignf(r, a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    q = fmaf(q, a, a);
    return t > 0.927734375f ? r : q;
}
// exact (erf) GELU, as torch.nn.functional.gelu(approximate="none") / HF ACT2FN["gelu"]
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erf_bf(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erf_bf(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Philox4x32-10 counter RNG (device-side draws for MLM masking / dropout).
#ifndef T4R_PHILOX_ROUNDS
#define T4R_PHILOX_ROUNDS 10
#endif
struct Philox {
    uint32_t k0, k1;
    __device__ __forceinline__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
    __device__ __forceinline__ uint4 operator()(uint64_t ctr_lo, uint64_t ctr_hi) const {
        uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32);
        uint32_t c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < T4R_PHILOX_ROUNDS; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
            const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ a;
            const uint32_t n1 = (uint32_t)p1;
            const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ b;
            const uint32_t n3 = (uint32_t)p0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
        return make_uint4(c0, c1, c2, c3);
    }
};
// The same block function in resumable form: a kernel that hides its Philox work in the shadow of matrix instructions
// evaluates it a round or two at a time (csrc/xlnet_fused.hip).  philox_begin + 10 x philox_round == Philox::operator().
struct PhiloxState { uint32_t c0, c1, c2, c3, a, b; };
__device__ __forceinline__ PhiloxState philox_begin(uint64_t seed, uint64_t ctr_lo, uint64_t ctr_hi) {
    PhiloxState s;
    s.c0 = (uint32_t)ctr_lo; s.c1 = (uint32_t)(ctr_lo >> 32); s.c2 = (uint32_t)ctr_hi; s.c3 = (uint32_t)(ctr_hi >> 32);
    s.a = (uint32_t)seed; s.b = (uint32_t)(seed >> 32);
    return s;
}
__device__ __forceinline__ void philox_round(PhiloxState& s) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * s.c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * s.c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ s.c1 ^ s.a;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ s.c3 ^ s.b;
    const uint32_t n3 = (uint32_t)p0;
    s.c0 = n0; s.c1 = n1; s.c2 = n2; s.c3 = n3;
    s.a += 0x9E3779B9u; s.b += 0xBB67AE85u;
}
__device__ __forceinline__ float u32_to_unit(uint32_t x) {  // [0,1)
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// ---------------------------------------------------------------------------------------------
// Dropout masks (torch.nn.Dropout semantics: keep with prob 1-p, scale kept values by 1/(1-p)).
// The keep decision of element `idx` at a dropout site is a pure function of
// (seed, ctr_hi, idx):  component (idx & 3) of Philox4x32-10(key = seed, counter = (idx >> 2, ctr_hi)).
// Forward and backward kernels recompute it; nothing is stored.  ctr_hi encodes
// (step offset << 16) | (layer << 8) | site, see xlnet_layer.hip.
struct DropCfg {
    float p;          // 0 => disabled
    float inv_keep;   // 1 / (1 - p)
    uint32_t thr;     // keep <=> word >= thr: the integer form of  u32_to_unit(word) >= p  (same decisions: the top 24
                      // bits k of the word give u = k 2^-24 exactly, so u >= p <=> k >= ceil(p 2^24) <=> word >= ceil(p 2^24) << 8);
                      // one compare per element instead of shift + convert + multiply + compare
    unsigned long long seed;
    unsigned long long ctr_hi;
};
__device__ __forceinline__ float drop_scale(const DropCfg& d, unsigned long long idx) {
    const Philox rng(d.seed);
    const uint4 r = rng(idx >> 2, d.ctr_hi);
    const unsigned k = (unsigned)(idx & 3);
    const uint32_t v = k == 0 ? r.x : (k == 1 ? r.y : (k == 2 ? r.z : r.w));
    return v >= d.thr ? d.inv_keep : 0.f;
}
// four consecutive elements starting at a multiple of 4
__device__ __forceinline__ float4 drop_scale4(const DropCfg& d, unsigned long long idx4) {
    const Philox rng(d.seed);
    const uint4 r = rng(idx4 >> 2, d.ctr_hi);
    return make_float4(r.x >= d.thr ? d.inv_keep : 0.f, r.y >= d.thr ? d.inv_keep : 0.f, r.z >= d.thr ? d.inv_keep : 0.f,
                       r.w >= d.thr ? d.inv_keep : 0.f);
}
// masks of the 16 consecutive elements idx0 .. idx0+15 of which the first n are needed (the rest get 1).
// aligned (wave uniform; caller guarantees idx0 % 4 == 0 and n % 4 == 0): one Philox block per four
// elements instead of one per element -- the block function is ~300 issue cycles per wave.
__device__ __forceinline__ void drop_scale_run16(const DropCfg& d, unsigned long long idx0, int n, bool aligned,
                                                 float (&m)[16]) {
    if (aligned) {
#pragma unroll
        for (int t = 0; t < 16; t += 4) {
            float4 f = make_float4(1.f, 1.f, 1.f, 1.f);
            if (t < n) f = drop_scale4(d, idx0 + t);
            m[t] = f.x; m[t + 1] = f.y; m[t + 2] = f.z; m[t + 3] = f.w;
        }
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) m[t] = t < n ? drop_scale(d, idx0 + t) : 1.f;
    }
}
// masks of VEC (1, 2 or 4) consecutive elements starting at idx0; aligned (wave uniform): idx0 % VEC == 0
// and the run does not straddle a Philox block, so one block serves the whole run
template <int VEC>
__device__ __forceinline__ void drop_scale_vec(const DropCfg& d, unsigned long long idx0, bool aligned, float (&m)[VEC]) {
    if (VEC > 1 && aligned) {
        const float4 f = drop_scale4(d, idx0 & ~3ull);
        if (VEC == 4) { m[0] = f.x; m[VEC > 1 ? 1 : 0] = f.y; m[VEC > 2 ? 2 : 0] = f.z; m[VEC > 3 ? 3 : 0] = f.w; }
        else { const bool hi = (idx0 & 2) != 0; m[0] = hi ? f.z : f.x; m[VEC > 1 ? 1 : 0] = hi ? f.w : f.y; }
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) m[e] = drop_scale(d, idx0 + e);
    }
}
// Masks for an MFMA accumulator fragment: a lane holds ONE column (col) of four consecutive rows
// row_base .. row_base+3, and the four lanes of a quad hold the four columns of one Philox block
// (row stride N % 4 == 0, col & ~3 aligned).  Lane q = col & 3 evaluates the block of row row_base + q
// and the quad transposes the 4 x 4 words with DPP broadcasts: one block function per lane instead of
// four.  All four lanes of the quad must be active.
template <int K>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, K * 0x55, 0xf, 0xf, true);
}
__device__ __forceinline__ void drop_scale_quad(const DropCfg& d, unsigned long long row_base, unsigned long long N,
                                                int col, float (&m)[4]) {
    const int q = col & 3;
    const Philox rng(d.seed);
    const uint4 w = rng(((row_base + q) * N + (unsigned long long)(col & ~3)) >> 2, d.ctr_hi);
#define T4R_QSEL(K)                                                                                      \
    {                                                                                                    \
        const uint32_t t0 = quad_bcast<K>(w.x), t1 = quad_bcast<K>(w.y), t2 = quad_bcast<K>(w.z),        \
                       t3 = quad_bcast<K>(w.w);                                                          \
        const uint32_t sel = q == 0 ? t0 : (q == 1 ? t1 : (q == 2 ? t2 : t3));                            \
        m[K] = sel >= d.thr ? d.inv_keep : 0.f;                                                           \
    }
    T4R_QSEL(0) T4R_QSEL(1) T4R_QSEL(2) T4R_QSEL(3)
#undef T4R_QSEL
}
static inline DropCfg make_drop(float p, unsigned long long seed, unsigned long long ctr_hi) {
    DropCfg d;
    d.p = p; d.inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f; d.seed = seed; d.ctr_hi = ctr_hi;
    d.thr = p <= 0.f ? 0u : (p >= 1.f ? 0xffffffffu : ((uint32_t)ceilf(p * 16777216.0f)) << 8);
    return d;
}
