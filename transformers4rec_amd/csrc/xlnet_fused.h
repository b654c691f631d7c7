// Shared device code of the token-tile-stationary kernels of the XLNet layer (xlnet_fused.hip: feed-forward block,
// xlnet_fused_attn.hip: the projections around the attention core): three-plane bf16 operands, the six-product MFMA
// chain, the weight-plane layout of a layer.  See xlnet_fused.hip for the design.
#pragma once
#include "t4r_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

int t4r_reduce_partials_launch(hipStream_t st, const float* part, int nblocks, float* o0, int n0, int a0,
                               float* o1, int n1, int a1, float* o2, int n2, int a2);   // elementwise.hip
// token blocks (of 16 rows) per workgroup tile of the token-tile kernels for T rows (xlnet_fused.hip).  backward = true: the
// launch may run next to a resident collective -- the CU budget of t4r_xlnet_set_cu_budget applies (forward launches never do)
int t4r_xlnet_pick_r(long T, bool backward);
bool t4r_xlnet_body_fp16x2();      // xlnet_fused.hip: the token-tile kernels on the two-way fp16 split (T4R_XLNET_FP16X2, default 1)?

__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ u32x4 ldq(const uint16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ f32x4 mfma_bf(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// two fp32 values -> one packed bf16 pair per plane (low half = first value), x = hi + mid + lo exactly; every cut rounds
// to nearest even (v_cvt_pk_bf16_f32) so that the dropped 2^-24 terms carry no systematic sign (gemm_kernel.h cvt_pair)
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    const f32x2v v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v));
}
__device__ __forceinline__ void cut3(float a, float b, uint32_t (&w)[3]) {
    w[0] = pk_bf16(a, b);
    const float ra = a - __uint_as_float(w[0] << 16), rb = b - __uint_as_float(w[0] & 0xffff0000u);
    w[1] = pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(w[1] << 16), sb = rb - __uint_as_float(w[1] & 0xffff0000u);
    w[2] = pk_bf16(sa, sb);
}
// the six partial products kept (planes 0 hi, 1 mid, 2 lo), smallest first
#define T4R_SIX(X) X(1, 1) X(2, 0) X(0, 2) X(1, 0) X(0, 1) X(0, 0)

// ------------------------------------------------------------------------------------------------ weight planes
// Every weight matrix of a layer is cut ONCE per layer call into three bf16 planes, stored in the orientation its product
// needs: an MFMA A fragment is 16 bytes of consecutive k of one output-feature row.  Matrices (rows = output feature of
// the product, cols = contraction index; planes [3][rows][cols], one after the other in this order):
//   QKVT [3D][D]   QKVT[z D + o][k] = W_z[k][o]      q, k, v projections (forward)
//   RT   [D][D]    RT[o][k] = r[k][o]                positional keys k_r (forward)
//   ON   [D][D]    o[h][nd] as stored               output projection (forward)
//   OT   [D][D]    OT[nd][h] = o[h][nd]              d attn_vec (backward)
//   QKVN [D][3D]   QKVN[k][z D + o] = W_z[k][o]      d h from d q, d k, d v (backward)
//   W1 [4D][D], W2 [D][4D] as stored (forward); W1T [D][4D], W2T [4D][D] transposed (backward)
struct LayerPlanes {
    const uint16_t *QKVT, *RT, *ON, *OT, *QKVN, *W1p, *W2p, *W1Tp, *W2Tp;
};
__host__ __device__ inline LayerPlanes carve_planes(const void* base, int D) {
    const uint16_t* b = (const uint16_t*)base;
    const long dd = 3L * D * D;
    LayerPlanes p;
    p.QKVT = b; b += 3 * dd;
    p.RT = b; b += dd;
    p.ON = b; b += dd;
    p.OT = b; b += dd;
    p.QKVN = b; b += 3 * dd;
    p.W1p = b; b += 4 * dd;
    p.W2p = b; b += 4 * dd;
    p.W1Tp = b; b += 4 * dd;
    p.W2Tp = b;
    return p;
}
// Two-way fp16 planes (x s = hi + lo, see csrc/head_split.hip: mfma_split): the same nine plane matrices once more, two
// planes each, every SOURCE matrix positioned by its own power-of-two scale (max |W| s in [2^13, 2^14)), then 16 floats:
//   scale[HS_Q, HS_K, HS_V, HS_R, HS_O, HS_W1, HS_W2] of the seven sources (both orientations of a source share it),
//   scale[HS_B1] = max |b1| (not a scale: with max |W1| = 2^14 / scale[HS_W1] rounded up it bounds |pre| per token).
struct LayerPlanesH {
    const uint16_t *QKVT, *RT, *ON, *OT, *QKVN, *W1p, *W2p, *W1Tp, *W2Tp;
    const float* scale;
};
enum { HS_Q = 0, HS_K = 1, HS_V = 2, HS_R = 3, HS_O = 4, HS_W1 = 5, HS_W2 = 6, HS_B1 = 7,
       HS_RESERVED = 8 };
// A workgroup's maximum of per-lane values -> out[blockIdx.x] (plain store: one slot per workgroup; the consumer takes the
// maximum over the slots.  Atomics on ONE word from 2 k waves cost 30-50 us per launch: same-address atomics serialise.)
// sh: NW floats of LDS nobody else uses any more; every thread of the workgroup calls it.
template <int NW>
__device__ __forceinline__ void block_amax_to(float m, float* sh, float* out, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        float mm = sh[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) mm = fmaxf(mm, sh[w]);
        out[blockIdx.x] = mm;
    }
}
static inline long layer_planes_bf16_floats(int D) { return 25L * 3 * D * D / 2; }
__host__ __device__ inline LayerPlanesH carve_planes_h(const void* base, int D) {
    const uint16_t* b = (const uint16_t*)((const float*)base + 25L * 3 * D * D / 2);
    const long dd = 2L * D * D;
    LayerPlanesH p;
    p.QKVT = b; b += 3 * dd;
    p.RT = b; b += dd;
    p.ON = b; b += dd;
    p.OT = b; b += dd;
    p.QKVN = b; b += 3 * dd;
    p.W1p = b; b += 4 * dd;
    p.W2p = b; b += 4 * dd;
    p.W1Tp = b; b += 4 * dd;
    p.W2Tp = b; b += 4 * dd;
    p.scale = (const float*)b;
    return p;
}
// Round 4: two fp32 TRANSPOSES behind the half-precision planes, for the exact-fp32 (v_mfma_f32_16x16x4_f32) attention
// block kernels (xlnet_attn_block.hip), whose A operand is 16 bytes of consecutive k of one output-feature row:
//   QKVT32 [3D][D]   QKVT32[z D + o][k] = W_z[k][o]     q, k, v projections (forward)
//   OT32   [D][D]    OT32[nd][h] = o[h][nd]             d attn_vec (backward)
// (o itself [h][nd] and W_z [k][o] as stored serve the o-projection and d h.)
static inline long layer_planes_half_floats(int D) { return layer_planes_bf16_floats(D) + 25L * 2 * D * D / 2 + 16; }
struct LayerPlanes32 { const float *QKVT, *OT; };
__host__ __device__ inline LayerPlanes32 carve_planes_32(const void* base, int D) {
    const float* b = (const float*)base + (25L * 3 * D * D / 2 + 25L * 2 * D * D / 2 + 16);
    LayerPlanes32 p;
    p.QKVT = b;
    p.OT = b + 3L * D * D;
    return p;
}
static inline long layer_planes_floats(int D) { return layer_planes_half_floats(D) + 4L * D * D; }

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 mfma_h(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// two (already scaled) fp32 values -> hi / lo fp16 pairs, both cuts round to nearest even.  Written as vector conversions and
// fma(hi, -1, x) so that the pair costs five vector instructions (v_cvt_pk_f16_f32, two v_cvt_f32_f16, v_pk_add_f32,
// v_cvt_pk_f16_f32) instead of eight (the element-wise form converted hi twice: once alone for the residual, once packed); same
// roundings, same bits.
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cut2h(float a, float b, uint32_t (&w)[2]) {
    const f32x2v ab = {a, b};
    const f16x2v h = __builtin_convertvector(ab, f16x2v);
    const f32x2v lab = {__builtin_fmaf((float)h[0], -1.0f, a), __builtin_fmaf((float)h[1], -1.0f, b)};
    const f16x2v l = __builtin_convertvector(lab, f16x2v);
    w[0] = __builtin_bit_cast(uint32_t, h);
    w[1] = __builtin_bit_cast(uint32_t, l);
}
// the power of two that puts a largest magnitude m into [2^13, 2^14) (1 for zero / non-finite m)
__device__ __forceinline__ float pow2_scale(float m) {
    if (!(m > 0.f) || !(m < 3e38f)) return 1.f;
    int e;
    (void)frexpf(m, &e);
    return ldexpf(1.f, min(14 - e, 100));    // a maximum below 2^-86 (the d W bound of an item far below every row's lse) must not overflow the scale
}

// acc[r] += A_tile (16 features x D) . B_r (D x 16 tokens) for every token block r, three-plane operands, six products.
// The weight fragment is requested one product ahead (load_a3: 3 D/32 16-byte loads from the L2-resident planes, issued
// before the previous product / epilogue so that their latency never sits in front of the matrix instructions).
// ap: plane 0 of the weight matrix at this lane's row and k offset 8 kg (row-major, k contiguous), apl: plane stride;
// bp: plane 0 of the LDS token planes at this lane's token row offset + 8 kg, bpl: plane stride, blocks 16 * P rows apart
template <int D>
struct AFrag { u32x4 v[D / 32][3]; };
#ifndef T4R_FF_PREFETCH
#define T4R_FF_PREFETCH 0   /* A/B, same box: fragments requested one product ahead cost 67 us per forward launch against 60 (237 VGPRs instead of 143) */
#endif
template <int D>
__device__ __forceinline__ void load_a3(AFrag<D>& a, const uint16_t* __restrict__ ap, long apl) {
#pragma unroll
    for (int s = 0; s < D / 32; ++s)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) a.v[s][pl] = ldq(ap + pl * apl + 32 * s);
}
template <int D, int R, int P>
__device__ __forceinline__ void product3(const AFrag<D>& a, const uint16_t* bp, int bpl, f32x4 (&acc)[R]) {
    constexpr int KS = D / 32;
    constexpr int RG = R > 3 ? 3 : R;          // token blocks per group: at most 3 x 3 planes of B fragments (36 VGPRs) live
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += RG) {
            u32x4 b[RG][3];
#pragma unroll
            for (int r = 0; r < RG; ++r)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    if (r0 + r < R) b[r][pl] = ldq(bp + pl * bpl + (r0 + r) * 16 * P + 32 * s);
#define T4R_PROD(PA, PB)                                                        \
            _Pragma("unroll") for (int r = 0; r < RG; ++r)                      \
                if (r0 + r < R) acc[r0 + r] = mfma_bf(a.v[s][PA], b[r][PB], acc[r0 + r]);
            T4R_SIX(T4R_PROD)
#undef T4R_PROD
        }
    }
}

// ---- the same product on two-way fp16 planes: three matrix instructions per k-step and token block instead of six
template <int D>
struct AFragH { u32x4 v[D / 32][2]; };
template <int D>
__device__ __forceinline__ void load_a2h(AFragH<D>& a, const uint16_t* __restrict__ ap, long apl) {
#pragma unroll
    for (int s = 0; s < D / 32; ++s)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) a.v[s][pl] = ldq(ap + pl * apl + 32 * s);
}
template <int D, int R, int P>
__device__ __forceinline__ void product3h(const AFragH<D>& a, const uint16_t* bp, int bpl, f32x4 (&acc)[R]) {
#pragma unroll
    for (int s = 0; s < D / 32; ++s) {
        u32x4 b[R][2];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) b[r][pl] = ldq(bp + pl * bpl + r * 16 * P + 32 * s);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = mfma_h(a.v[s][1], b[r][0], acc[r]);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = mfma_h(a.v[s][0], b[r][1], acc[r]);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = mfma_h(a.v[s][0], b[r][0], acc[r]);
    }
}
