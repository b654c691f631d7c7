// Next-item head (tied full softmax) for d_model <= 128, fp32-class accuracy on the 16-bit matrix cores.
//
// Two operand forms live here.  Round 2 (described first, still selectable with T4R_HEAD_FWD_FP16X2=0 /
// T4R_HEAD_DW_FP16X2=0): three bf16 planes per operand, six products.  Round 3 (default, see mfma_split below): a two-way
// fp16 split with exact power-of-two scales -- one per tensor for the forward and d X, one per ITEM for d W -- three
// products; same or smaller error against fp64, half the matrix instructions (these kernels run at the package power limit).
//
// Replaces, for D = 32 / 64 / 96 / 128, the three vocabulary-wide contractions of
// transformers4rec/torch/model/prediction_task.py:664 (logits = X @ W^T) and of its autograd
// (d X = dlogits @ W, d W = dlogits^T @ X, with CrossEntropyLoss' backward :446 formed on the fly from the
// stored logits) -- 2.1 of the 4.8 ms training step at BASELINE configs[1] when they ran through the general
// GEMM (gemm_kernel.h, PREC 1).  Same arithmetic as PREC 1: every operand is cut into three bf16 pieces
// (x = hi + mid + lo exactly, round-to-nearest cuts) and the six largest partial products are accumulated in
// fp32 by v_mfma_f32_32x32x16_bf16.  What changes is WHERE the cutting happens: the general kernel re-cuts every
// operand tile each time a workgroup stages it (the VALU work equals the matrix-core time), here
//   * X [N, D] (a few thousand label rows) is cut ONCE per step into fragment-ordered plane blocks
//     (split_mk / split_km kernels: 24 KB per 32 rows);
//   * logits: a workgroup keeps its 128 rows of W as MFMA B fragments IN REGISTERS for its whole life (cut once per
//     workgroup, the whole K = D extent: 96 VGPRs at D = 128) and streams the X plane blocks through LDS --
//     no conversion and one ds_read_b128 per two MFMAs in the loop;
//   * d W: the softmax-gradient fragment G^T (vocabulary x rows) is computed by the lane that feeds it to the
//     matrix core, straight from coalesced logit loads -- every element of the [N, V] gradient is formed and cut
//     exactly once in the whole grid and never touches LDS; X arrives as pre-cut plane blocks through LDS;
//   * d X: the same scheme with rows as M (each lane walks its own logits row), W^T as pre-cut plane blocks
//     (split_km over the table, once per step), split over the vocabulary into deterministic partial sums
//     (no atomics) that a small kernel adds in a fixed order.
// MFMA operand convention used by all kernels here: the k-slot (lane >> 5, i) of step s holds physical
// k = 16 s + 8 (lane >> 5) + i, i = position in the lane's 16 bytes -- the same map for A and B, so any
// consistent order is a valid contraction order.
#include "gemm_kernel.h"

namespace {

constexpr float kLog2e = 1.4426950408889634f;
// native vector type for everything that is staged: assigning HIP's uint4 struct between address spaces becomes a
// memcpy that keeps the staging array in scratch memory
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

__device__ __forceinline__ void split8(const float (&x)[8], u32x4 (&w)[3]) {
    uint32_t a[3], b[3], c[3], d[3];
    cvt_pair<1>(x[0], x[1], a);
    cvt_pair<1>(x[2], x[3], b);
    cvt_pair<1>(x[4], x[5], c);
    cvt_pair<1>(x[6], x[7], d);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) w[pl] = u32x4{a[pl], b[pl], c[pl], d[pl]};
}
// the six partial products of one K = 16 step, smallest first
__device__ __forceinline__ f32x16 mfma6(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 acc) {
    acc = mfma_bf16(a[1], b[1], acc);
    acc = mfma_bf16(a[2], b[0], acc);
    acc = mfma_bf16(a[0], b[2], acc);
    acc = mfma_bf16(a[1], b[0], acc);
    acc = mfma_bf16(a[0], b[1], acc);
    acc = mfma_bf16(a[0], b[0], acc);
    return acc;
}

// ---- the forward products in the TWO-way fp16 split (HS = true).  The head's three kernels run at the 1.4 kW package
// limit with the clock pulled down (tools/head_power_probe.py): what they cost is matrix instructions.  Where BOTH operands
// are well scaled -- the forward product: rows of a LayerNorm output x the item table -- a two-way fp16 split needs three
// products instead of six:  x s = hi + lo + d,  hi = fp16(x s), lo = fp16(x s - hi), |d| <= 2^-23 |x s|, with s a power of two
// that puts the tensor's largest magnitude into [2^13, 2^14) (exact; undone exactly in the epilogue), products
// hi hi + hi lo + lo hi (the dropped lo lo <= 2^-22 |x w|): per product term <= 2^-21 |x w| in the worst case, measured
// 2-3 x the error of the fp32 matrix cores against fp64 (tools/head_split_bench.py).  Entries below 2^-17 of the
// tensor's maximum keep fewer than 22 bits (fp16 subnormals in lo) -- irrelevant for a dot product's norm-wise error, NOT
// acceptable for the gradient operand, whose rare-item columns lie 10^-6 below its maximum: the backward products stay
// on the three bf16 planes.  T4R_HEAD_FWD_FP16X2=0 puts the forward back on them too.
__device__ __forceinline__ f32x16 mfma_f16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
}
// (vector conversions + fma(hi, -1, x): five vector instructions per pair instead of eight, same roundings -- xlnet_fused.h: cut2h)
typedef float float2v_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cut_pair_h(float a, float b, uint32_t (&w)[2]) {
    const float2v_t ab = {a, b};
    const half2_t h = __builtin_convertvector(ab, half2_t);     // round to nearest even
    const float2v_t lab = {__builtin_fmaf((float)h[0], -1.0f, a), __builtin_fmaf((float)h[1], -1.0f, b)};
    const half2_t l = __builtin_convertvector(lab, half2_t);
    w[0] = __builtin_bit_cast(uint32_t, h);
    w[1] = __builtin_bit_cast(uint32_t, l);
}
template <bool HS>
__device__ __forceinline__ void split8s(const float (&x)[8], float scale, u32x4 (&w)[HS ? 2 : 3]) {
    if constexpr (HS) {
        uint32_t a[2], b[2], c[2], d[2];
        cut_pair_h(x[0] * scale, x[1] * scale, a);
        cut_pair_h(x[2] * scale, x[3] * scale, b);
        cut_pair_h(x[4] * scale, x[5] * scale, c);
        cut_pair_h(x[6] * scale, x[7] * scale, d);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) w[pl] = u32x4{a[pl], b[pl], c[pl], d[pl]};
    } else {
        split8(x, w);
    }
}
template <bool HS>
__device__ __forceinline__ f32x16 mfma_split(const u32x4 (&a)[HS ? 2 : 3], const u32x4 (&b)[HS ? 2 : 3], f32x16 acc) {
    if constexpr (HS) {
        acc = mfma_f16(a[1], b[0], acc);
        acc = mfma_f16(a[0], b[1], acc);
        acc = mfma_f16(a[0], b[0], acc);
        return acc;
    } else {
        return mfma6(a, b, acc);
    }
}
// the same three products with the operand ROLES exchanged (a is what the other kernel passes as b): the terms are added
// in the same order, so a score tile recomputed with X as the A operand equals the forward's (W as the A operand) bit for bit
__device__ __forceinline__ f32x16 mfma_split_swapped(const u32x4 (&a)[2], const u32x4 (&b)[2], f32x16 acc) {
    acc = mfma_f16(a[0], b[1], acc);
    acc = mfma_f16(a[1], b[0], acc);
    acc = mfma_f16(a[0], b[0], acc);
    return acc;
}
// largest magnitude of a [rows, cols] matrix as the bits of a non-negative float (they order like unsigned integers), and
// the power of two that maps it into [2^13, 2^14)
__global__ __launch_bounds__(1024) void amax_kernel(const float* __restrict__ src, long ld, long rows, int cols,
                                                     unsigned* __restrict__ out) {
    const long n4 = rows * (cols / 4), stride = (long)gridDim.x * blockDim.x;
    const int c4n = cols / 4;
    float m = 0.f;
    // four requests in flight per lane (clamped addresses: a repeated element does not change a maximum)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long j = min(i + u * stride, n4 - 1);
            v[u] = *reinterpret_cast<const float4*>(src + (j / c4n) * ld + (int)(j % c4n) * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    // ONE atomic per workgroup and at most 256 workgroups (of up to 1024 threads): atomics on one word serialise at the memory
    // side at ~12 ns each -- one per wave of 2048 workgroups made the launch 75-100 us, one per workgroup of 1024 still 27 us
    // for the 51 MB table (12 us of it the atomics; round 4)
    __shared__ float red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (int)(blockDim.x >> 6);
        for (int w = 1; w < nw; ++w) m = fmaxf(m, red[w]);
        if (m > 0.f) atomicMax(out, __float_as_uint(m));
    }
}
__device__ __forceinline__ float scale_of(const unsigned* amax_bits) {
    const float a = __uint_as_float(*amax_bits);
    if (!(a > 0.f) || !(a < 3e38f)) return 1.f;      // all zero, infinite or NaN: nothing to position
    int e;
    (void)frexpf(a, &e);                              // a = m 2^e, m in [0.5, 1)
    return ldexpf(1.f, min(14 - e, 100));    // a maximum below 2^-86 (the d W bound of an item far below every row's lse) must not overflow the scale
}

// ---- plane blocks.  One block = 32 rows of a row-major fp32 matrix [n_rows, D], as three bf16 planes:
//   MK image (rows are the M / N index of the product, D is K):  [plane][chunk = d / 8][row % 32] x 16 bytes = 8 consecutive d
//   KM image (rows are K, D is the N index):                     [plane][kc = (row % 32) / 8][d] x 16 bytes = 8 consecutive rows
// Both are 12 D u32x4 (24 KB at D = 128); rows >= n_rows are zero.
template <int NB, bool HS = false>
__device__ __forceinline__ void split_mk_body(int b, const float* __restrict__ src, long ld, int n_rows,
                                              u32x4* __restrict__ dst, const unsigned* __restrict__ amax) {
    constexpr int D = 32 * NB, CH = D / 8, NPL = HS ? 2 : 3;
    const float scale = HS ? scale_of(amax) : 1.f;
    for (int idx = threadIdx.x; idx < CH * 32; idx += 256) {
        const int r = idx & 31, c = idx >> 5, row = b * 32 + r;
        float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (row < n_rows) {
            const float4 u = *reinterpret_cast<const float4*>(src + (long)row * ld + 8 * c);
            const float4 v = *reinterpret_cast<const float4*>(src + (long)row * ld + 8 * c + 4);
            x[0] = u.x; x[1] = u.y; x[2] = u.z; x[3] = u.w; x[4] = v.x; x[5] = v.y; x[6] = v.z; x[7] = v.w;
        }
        u32x4 w[NPL];
        split8s<HS>(x, scale, w);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) dst[((long)b * NPL + pl) * (CH * 32) + c * 32 + r] = w[pl];
    }
}
template <int NB, bool HS = false>
__global__ __launch_bounds__(256) void split_mk_kernel(const float* __restrict__ src, long ld, int n_rows,
                                                        u32x4* __restrict__ dst, const unsigned* __restrict__ amax = nullptr) {
    split_mk_body<NB, HS>(blockIdx.x, src, ld, n_rows, dst, amax);
}
template <int NB, bool HS = false>
__device__ __forceinline__ void split_km_body(int b, const float* __restrict__ src, long ld, int n_rows,
                                              u32x4* __restrict__ dst, const unsigned* __restrict__ amax) {
    constexpr int D = 32 * NB, NPL = HS ? 2 : 3;
    const float scale = HS ? scale_of(amax) : 1.f;
    for (int idx = threadIdx.x; idx < 4 * D; idx += 256) {
        const int d = idx % D, kc = idx / D;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int row = b * 32 + 8 * kc + e;
            x[e] = row < n_rows ? src[(long)row * ld + d] : 0.f;
        }
        u32x4 w[NPL];
        split8s<HS>(x, scale, w);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) dst[(((long)b * NPL + pl) * 4 + kc) * D + d] = w[pl];
    }
}
template <int NB, bool HS = false>
__global__ __launch_bounds__(256) void split_km_kernel(const float* __restrict__ src, long ld, int n_rows,
                                                        u32x4* __restrict__ dst, const unsigned* __restrict__ amax = nullptr) {
    split_km_body<NB, HS>(blockIdx.x, src, ld, n_rows, dst, amax);
}
// the three images of the head's input rows in ONE launch (blockIdx.y picks the image): fp16 MK (logits), bf16 KM and fp16 KM
// (d W in either form) -- three 5 us launches on the critical stream otherwise
template <int NB>
__global__ __launch_bounds__(256) void split_x_images_kernel(const float* __restrict__ src, long ld, int n_rows, u32x4* __restrict__ xa,
                                                              u32x4* __restrict__ xt, u32x4* __restrict__ xth,
                                                              const unsigned* __restrict__ amax) {
    if (blockIdx.y == 0) split_mk_body<NB, true>(blockIdx.x, src, ld, n_rows, xa, amax);
    else if (blockIdx.y == 1) split_km_body<NB, false>(blockIdx.x, src, ld, n_rows, xt, nullptr);
    else split_km_body<NB, true>(blockIdx.x, src, ld, n_rows, xth, amax);
}

// KMP image (round 4, the recomputing backward kernels): as the KM image, with the 32 rows of a block in the order the
// 32 x 32 ACCUMULATOR hands them to a lane -- position (kc = 2 s + khalf, e) holds row 16 s + 4 khalf + (e & 3) + 8 (e >> 2),
// i.e. accumulator register 8 s + e of lane half khalf.  A score tile recomputed on the matrix cores then IS (after the
// softmax-gradient transform, in registers) the A operand of the product that contracts over its rows, against this image.
template <int NB, bool HS = false>
__global__ __launch_bounds__(256) void split_kmp_kernel(const float* __restrict__ src, long ld, int n_rows,
                                                         u32x4* __restrict__ dst, const unsigned* __restrict__ amax = nullptr) {
    constexpr int D = 32 * NB, NPL = HS ? 2 : 3;
    const int b = blockIdx.x;
    const float scale = HS ? scale_of(amax) : 1.f;
    for (int idx = threadIdx.x; idx < 4 * D; idx += 256) {
        const int d = idx % D, kc = idx / D;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int row = b * 32 + 16 * (kc >> 1) + 4 * (kc & 1) + (e & 3) + 8 * (e >> 2);
            x[e] = row < n_rows ? src[(long)row * ld + d] : 0.f;
        }
        u32x4 w[NPL];
        split8s<HS>(x, scale, w);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) dst[(((long)b * NPL + pl) * 4 + kc) * D + d] = w[pl];
    }
}

// ---- logits:  C[N, V] = alpha * X @ W^T.  grid (ceil(V / 128), row splits); X as MK plane blocks.
template <int NB, bool HS = false>
__global__ __launch_bounds__(256) void head_logits_split_kernel(const u32x4* __restrict__ XA, const float* __restrict__ W,
                                                                 long ldw, float* __restrict__ C, long ldc, int N, int V,
                                                                 float alpha, int nblk, int blk_per,
                                                                 const unsigned* __restrict__ amax = nullptr) {
    constexpr int KS = 2 * NB, CH = 4 * NB, NPL = HS ? 2 : 3;
    constexpr int BLK = 4 * NPL * 32 * NB;      // u32x4 per plane block
    const float sw = HS ? scale_of(amax + 1) : 1.f;
    if (HS) alpha = (alpha / scale_of(amax)) / sw;      // two exact divisions by powers of two (their product may overflow)
    constexpr int SN = (BLK + 255) / 256;
    __shared__ u32x4 lds[2][BLK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, khalf = lane >> 5;
    const int b_begin = blockIdx.y * blk_per, b_end = min(nblk, b_begin + blk_per);
    if (b_begin >= b_end) return;
    const int col = blockIdx.x * 128 + 32 * wave + l32;

    // this lane's 8-element pieces of row `col` of W, cut once
    u32x4 Bf[KS][NPL];
    {
        const float* wr = W + (long)min(col, V - 1) * ldw + 8 * khalf;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 u = *reinterpret_cast<const float4*>(wr + 16 * s);
            const float4 v = *reinterpret_cast<const float4*>(wr + 16 * s + 4);
            const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
            split8s<HS>(x, sw, Bf[s]);
        }
    }

    u32x4 st[SN];
    auto g_load = [&](int b) __attribute__((always_inline)) {
        const u32x4* src = XA + (long)b * BLK;
#pragma unroll
        for (int i = 0; i < SN; ++i)
            if (BLK % 256 == 0 || i * 256 + tid < BLK) st[i] = src[i * 256 + tid];
    };
    auto s_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < SN; ++i)
            if (BLK % 256 == 0 || i * 256 + tid < BLK) lds[buf][i * 256 + tid] = st[i];
    };
    g_load(b_begin);
    s_store(0);
    __syncthreads();
    for (int b = b_begin; b < b_end; ++b) {
        const int buf = (b - b_begin) & 1;
        // unconditional (the last pass re-reads its own block); pinned BEFORE the MFMAs -- left alone the scheduler sinks
        // the loads below them and the full L2 latency is exposed at the LDS stores
        g_load(min(b + 1, b_end - 1));
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 a[NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) a[pl] = lds[buf][(pl * CH + 2 * s + khalf) * 32 + l32];
            acc = mfma_split<HS>(a, Bf[s], acc);
        }
        // next block -> LDS before the logits are stored: the wait for its loads must not cover the HBM stores
        s_store(buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if (col < V) {
            const int row0 = b * 32 + 4 * khalf;
            float* c0 = C + (long)row0 * ldc + col;
            if (b * 32 + 32 <= N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) c0[(long)((r & 3) + 8 * (r >> 2)) * ldc] = alpha * acc[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (row0 + dr < N) c0[(long)dr * ldc] = alpha * acc[r];
                }
            }
        }
        __syncthreads();
    }
}

// ---- logits + cross-entropy statistics in one pass.  Same product with the operand roles swapped (W fragments as the
// A operand): the accumulator of lane (l32, khalf) then holds SIXTEEN VOCABULARY columns of ONE row of X -- the row's
// (max, sum-exp, sum) over them are lane-local reductions, the lane pair (khalf 0 / 1) and the four waves merge
// through one shuffle and 2 KB of LDS, and each workgroup leaves one partial triple per (row, 128-column tile):
// softmax_ce_fwd's second pass over the 1.1 GB of logits (0.24 ms at BASELINE configs[1]) becomes a 26 MB merge.
// Stores: a lane owns four consecutive columns per accumulator quad -> one 16-byte store each.
__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
    const float mn = fmaxf(m, m2);
    const float ref = mn == -INFINITY ? 0.f : mn;
    s = s * __builtin_amdgcn_exp2f((m - ref) * kLog2e) + s2 * __builtin_amdgcn_exp2f((m2 - ref) * kLog2e);
    m = mn;
}
// 4 x 4 transpose between the four lanes of a quad and four registers: on return x[g] holds what lane (quad base + g)
// had in x[t], t = this lane's position in the quad (two DPP exchange steps).
__device__ __forceinline__ void quad_transpose4(float (&x)[4], int t) {
    // Written so that every DPP move has ONE consumer, a select: the compiler folds the pair into v_cndmask_b32_dpp (8 vector
    // instructions per transpose; the send-select / move / receive-select form took 16).  Exchange with lane t ^ 1: an even lane's
    // x[1] becomes its partner's x[0], an odd lane's x[0] its partner's x[1] (and the same for x[3] / x[2]); then with lane t ^ 2 on
    // the register pairs (0, 2) and (1, 3).
    const bool odd = t & 1, hi = t & 2;
    auto swap1 = [](float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true)); };
    auto swap2 = [](float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true)); };
    {   // (every move is executed by ALL lanes, outside the selects: a cross-lane move under a divergent condition reads disabled lanes)
        const float a0 = x[0], a1 = x[1], a2 = x[2], a3 = x[3];
        const float p0 = swap1(a0), p1 = swap1(a1), p2 = swap1(a2), p3 = swap1(a3);
        x[1] = odd ? a1 : p0;
        x[0] = odd ? p1 : a0;
        x[3] = odd ? a3 : p2;
        x[2] = odd ? p3 : a2;
    }
    {
        const float a0 = x[0], a1 = x[1], a2 = x[2], a3 = x[3];
        const float p0 = swap2(a0), p1 = swap2(a1), p2 = swap2(a2), p3 = swap2(a3);
        x[2] = hi ? a2 : p0;
        x[0] = hi ? p2 : a0;
        x[3] = hi ? a3 : p1;
        x[1] = hi ? p3 : a1;
    }
}

template <int NB, bool HS, bool ZL = false>
__device__ __forceinline__ void head_logits_ce_body(const u32x4* __restrict__ XA, const float* __restrict__ W,
                                                    long ldw, float* __restrict__ C, long ldc, int N, int V,
                                                    float alpha, int nblk, int blk_per, int vec_ok,
                                                    float* __restrict__ st_m, float* __restrict__ st_s,
                                                    float* __restrict__ st_t, int n_tile, int n_split,
                                                    const unsigned* __restrict__ amax, float* __restrict__ colmax, int vpad,
                                                    const long* __restrict__ labels = nullptr, float* __restrict__ zlab = nullptr) {
    // C == NULL (the recomputing head, round 4): nothing is stored but the statistics -- and, with `labels`, each row's
    // label logit (zlab [N]: the loss needs it, and the lane that holds it writes it: exact, no second product)
    constexpr int KS = 2 * NB, CH = 4 * NB, NPL = HS ? 2 : 3;
    constexpr int BLK = 4 * NPL * 32 * NB;
    const float sw = HS ? scale_of(amax + 1) : 1.f;
    if (HS) alpha = (alpha / scale_of(amax)) / sw;      // two exact divisions by powers of two (their product may overflow)
    constexpr int SN = (BLK + 255) / 256;
    __shared__ u32x4 lds[2][BLK];
    __shared__ float4 sst[2][4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, khalf = lane >> 5;
    // XCD-aware order (workgroup b runs on XCD b % 8): the row splits of ONE column tile sit on one XCD, eight
    // dispatch slots apart, so its 64 KB of W come into that L2 once (row-major order: FETCH_SIZE 450 MB per launch,
    // every split re-read its tile through the fabric)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = (slot / n_split) * 8 + xcd, rsi = slot % n_split;
    if (tile >= n_tile) return;
    const int b_begin = rsi * blk_per, b_end = min(nblk, b_begin + blk_per);
    if (b_begin >= b_end) return;
    const int vbase = tile * 128 + 32 * wave;
    const bool tail_tile = tile * 128 + 128 > V;

    u32x4 Wf[KS][NPL];
    {
        const float* wr = W + (long)min(vbase + l32, V - 1) * ldw + 8 * khalf;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 u = *reinterpret_cast<const float4*>(wr + 16 * s);
            const float4 v = *reinterpret_cast<const float4*>(wr + 16 * s + 4);
            const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
            split8s<HS>(x, sw, Wf[s]);
        }
    }
    u32x4 st[SN];
    long lab_n = 0;           // ZL: the label of this lane's row in the block being fetched (unconditional load, as the rest)
    auto g_load = [&](int b) __attribute__((always_inline)) {
        // (the label first: its consumer, at the top of the next pass, then waits for this load only, not for the block behind it)
        if constexpr (ZL) lab_n = labels[min(b * 32 + l32, N - 1)];
        const u32x4* src = XA + (long)b * BLK;
#pragma unroll
        for (int i = 0; i < SN; ++i)
            if (BLK % 256 == 0 || i * 256 + tid < BLK) st[i] = src[i * 256 + tid];
    };
    auto s_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < SN; ++i)
            if (BLK % 256 == 0 || i * 256 + tid < BLK) lds[buf][i * 256 + tid] = st[i];
    };
    // the four waves' partials of one row block -> one triple per row (threads 0..31, after the block's barrier)
    auto flush_stats = [&](int buf, int b) __attribute__((always_inline)) {
        if (tid < 32) {
            float4 a = sst[buf][0][tid];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 o = sst[buf][w][tid];
                lse_merge(a.x, a.y, o.x, o.y);
                a.z += o.z;
            }
            const int row = b * 32 + tid;
            if (row < N) {
                const long o = (long)tile * N + row;
                st_m[o] = a.x;
                st_s[o] = a.y;
                if (st_t) st_t[o] = a.z;
            }
        }
    };
    // running maxima of this lane's sixteen columns over the rows of this workgroup (d W's per-item scales: head_dw_split_kernel)
    float cmax[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) cmax[r] = -INFINITY;
    g_load(b_begin);
    s_store(0);
    __syncthreads();
    for (int b = b_begin; b < b_end; ++b) {
        const int buf = (b - b_begin) & 1;
        const long lab_c = lab_n;
        g_load(min(b + 1, b_end - 1));
        __builtin_amdgcn_sched_barrier(0);
        if (b > b_begin) flush_stats(buf ^ 1, b - 1);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 xf[NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) xf[pl] = lds[buf][(pl * CH + 2 * s + khalf) * 32 + l32];
            acc = mfma_split<HS>(Wf[s], xf, acc);
        }
        s_store(buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        // lane (l32, khalf): row b*32 + l32 of X, columns vbase + 8 g + 4 khalf + (0..3), g = 0..3
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = alpha * acc[r];
        const int row = b * 32 + l32;
        const int c0 = vbase + 4 * khalf;
        if (colmax && row < N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) cmax[r] = fmaxf(cmax[r], v[r]);
        }
        if (ZL && row < N) {
            const int o = (int)(lab_c - c0);                 // this lane's columns: c0 + 8 g + i, g, i = 0..3
            if (o >= 0 && o < 32 && (o & 4) == 0) {
                const int rsel = 4 * (o >> 3) + (o & 3);
                float zl = v[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) zl = rsel == r ? v[r] : zl;
                zlab[row] = zl;
            }
        }
        if (ZL) {
        } else if (vec_ok && !tail_tile) {
            // full-line stores: the quad's four rows x four column groups are transposed through DPP, so that one store
            // instruction covers EIGHT lanes = all 128 bytes of a row (storing the accumulator layout as it is writes
            // 32 bytes per row and instruction: FETCH_SIZE showed a third of the logits lines read back for merging)
            const int t = l32 & 3;
            float w[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x[4] = {v[i], v[4 + i], v[8 + i], v[12 + i]};
                quad_transpose4(x, t);
#pragma unroll
                for (int g = 0; g < 4; ++g) w[g][i] = x[g];
            }
            const int rq = b * 32 + (l32 & ~3);
            float* cq = C + (long)rq * ldc + vbase + 8 * t + 4 * khalf;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (rq + g < N) {
                    // streaming store: the 1.1 GB of logits must not push the X plane blocks out of L2
                    const f32x4 o = {w[g][0], w[g][1], w[g][2], w[g][3]};
                    __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(cq + (long)g * ldc));
                }
        } else if (row < N) {
            float* cr = C + (long)row * ldc + c0;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (c0 + 8 * (r >> 2) + (r & 3) < V) cr[8 * (r >> 2) + (r & 3)] = v[r];
        }
        float t = 0.f;
        if (tail_tile) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool in = c0 + 8 * (r >> 2) + (r & 3) < V;
                t += in ? v[r] : 0.f;
                v[r] = in ? v[r] : -INFINITY;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) t += v[r];
        }
        float m = v[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fmaxf(m, v[r]);
        const float ref = (m == -INFINITY ? 0.f : m) * kLog2e;
        float se = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) se += __builtin_amdgcn_exp2f(fmaf(v[r], kLog2e, -ref));
        lse_merge(m, se, __shfl_xor(m, 32, 64), __shfl_xor(se, 32, 64));
        t += __shfl_xor(t, 32, 64);
        if (khalf == 0) sst[buf][wave][l32] = make_float4(m, se, t, 0.f);
        __syncthreads();
    }
    flush_stats((b_end - 1 - b_begin) & 1, b_end - 1);
    if (colmax) {       // one reduction over the 32 rows of the wave per workgroup, not per row block
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) cmax[r] = fmaxf(cmax[r], __shfl_xor(cmax[r], o, 64));
        }
        if (l32 == 0) {
            float* cp = colmax + (long)rsi * vpad + vbase + 4 * khalf;       // columns vbase + 8 g + 4 khalf + (0..3)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(cp + 8 * g) = make_float4(cmax[4 * g], cmax[4 * g + 1], cmax[4 * g + 2], cmax[4 * g + 3]);
        }
    }
}

// The fp16 form is pinned to three waves per SIMD (158 VGPRs, no spills): with the sixteen running column maxima it sat at 172,
// one register granule above the third wave, and the launch lost 15 %.  The bf16 form (212 VGPRs) keeps its two.
template <int NB, bool HS = false>
__global__ __launch_bounds__(256) void head_logits_ce_kernel(const u32x4* __restrict__ XA, const float* __restrict__ W,
                                                              long ldw, float* __restrict__ C, long ldc, int N, int V,
                                                              float alpha, int nblk, int blk_per, int vec_ok,
                                                              float* __restrict__ st_m, float* __restrict__ st_s,
                                                              float* __restrict__ st_t, int n_tile, int n_split,
                                                              const unsigned* __restrict__ amax = nullptr,
                                                              float* __restrict__ colmax = nullptr, int vpad = 0) {
    head_logits_ce_body<NB, false>(XA, W, ldw, C, ldc, N, V, alpha, nblk, blk_per, vec_ok, st_m, st_s, st_t, n_tile, n_split, amax,
                                   colmax, vpad);
}
template <int NB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void head_logits_ce_h_kernel(
    const u32x4* __restrict__ XA, const float* __restrict__ W, long ldw, float* __restrict__ C, long ldc, int N, int V, float alpha,
    int nblk, int blk_per, int vec_ok, float* __restrict__ st_m, float* __restrict__ st_s, float* __restrict__ st_t, int n_tile,
    int n_split, const unsigned* __restrict__ amax, float* __restrict__ colmax, int vpad) {
    head_logits_ce_body<NB, true>(XA, W, ldw, C, ldc, N, V, alpha, nblk, blk_per, vec_ok, st_m, st_s, st_t, n_tile, n_split, amax,
                                  colmax, vpad);
}
// the recomputing head's forward: statistics, column maxima and the label logits only -- no [N, V] tensor is written
template <int NB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void head_ce_stats_h_kernel(
    const u32x4* __restrict__ XA, const float* __restrict__ W, long ldw, int N, int V, float alpha, int nblk, int blk_per,
    float* __restrict__ st_m, float* __restrict__ st_s, float* __restrict__ st_t, int n_tile, int n_split,
    const unsigned* __restrict__ amax, float* __restrict__ colmax, int vpad, const long* __restrict__ labels,
    float* __restrict__ zlab) {
    head_logits_ce_body<NB, true, true>(XA, W, ldw, nullptr, 0, N, V, alpha, nblk, blk_per, 1, st_m, st_s, st_t, n_tile, n_split, amax,
                                        colmax, vpad, labels, zlab);
}

// d W's per-item scales need: min over the rows of lse, and which items are some row's label (their column holds a -g (1 - eps))
__global__ __launch_bounds__(1024) void head_dw_aux_kernel(const float* __restrict__ lse, const long* __restrict__ labels, int N,
                                                            int yoff, int Vc, float* __restrict__ lse_min,
                                                            unsigned char* __restrict__ islab, int vpad) {
    __shared__ float red[16];
    // the flags are cleared here (one workgroup: a barrier orders the clear before the marks) -- a hipMemsetAsync of the
    // ~100 KB array in front of this launch became three fill kernels of ~6 us each on the critical stream
    for (int i = threadIdx.x; i < vpad / 16; i += 1024) reinterpret_cast<uint4*>(islab)[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = (vpad / 16) * 16 + threadIdx.x; i < vpad; i += 1024) islab[i] = 0;
    __threadfence_block();
    __syncthreads();
    float m = INFINITY;
    for (int i = threadIdx.x; i < N; i += 1024) {
        m = fminf(m, lse[i]);
        const long y = labels[i] - yoff;
        if (y >= 0 && y < Vc) islab[y] = 1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = fminf(m, red[w]);
        *lse_min = fminf(m, red[0]);
    }
}

// per row: merge the column tiles' partial triples -> lse, loss  (32 rows x 32 tile slices per workgroup: a row's ~800
// partials are read by 32 threads, adjacent threads read adjacent rows)
//   loss_row = lse - (1 - eps) * logit[y] - eps / V * sum_j logit[j]      (torch CrossEntropyLoss, label smoothing eps)
__global__ __launch_bounds__(1024) void head_ce_finalize_kernel(const float* __restrict__ st_m, const float* __restrict__ st_s,
                                                                 const float* __restrict__ st_t, int n_tiles, int N, int V,
                                                                 const float* __restrict__ C, long ldc,
                                                                 const long* __restrict__ labels, float smoothing,
                                                                 float* __restrict__ loss_rows, float* __restrict__ lse_out,
                                                                 const float* __restrict__ zlab = nullptr) {
    __shared__ float sm[32][33], ss[32][33], stt[32][33];
    const int r = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int row = min(blockIdx.x * 32 + r, N - 1);
    float m = -INFINITY, s = 0.f, t = 0.f;
    for (int k = sl; k < n_tiles; k += 32) {
        const long o = (long)k * N + row;
        lse_merge(m, s, st_m[o], st_s[o]);
        if (st_t) t += st_t[o];
    }
    sm[sl][r] = m; ss[sl][r] = s; stt[sl][r] = t;
    __syncthreads();
    if (sl == 0 && blockIdx.x * 32 + r < N) {
        for (int k = 1; k < 32; ++k) { lse_merge(m, s, sm[k][r], ss[k][r]); t += stt[k][r]; }
        const float lse = m + __logf(s);
        float loss = lse - (zlab ? zlab[row] : C[(long)row * ldc + labels[row]]);
        if (smoothing > 0.f) loss = (1.f - smoothing) * loss + smoothing * (lse - t / V);
        loss_rows[row] = loss;
        lse_out[row] = lse;
    }
}

// softmax-CE gradient of one logit:  g * (exp(x - lse) - eps / V) - [col == y] * g * (1 - eps),  l2 = lse * log2(e)
struct SgScalars { float g, sub, hit; };
__device__ __forceinline__ float sg_value(float x, float l2, bool is_label, const SgScalars& q) {
    const float t = __builtin_amdgcn_exp2f(fmaf(x, kLog2e, -l2));
    return fmaf(q.g, t, -q.sub) - (is_label ? q.hit : 0.f);
}

// ---- d W[Vc, D] (+)= alpha * G^T @ X.  grid ceil(Vc / 128); wave w owns vocabulary rows 32 w .. 32 w + 31 of the tile.
// HS (T4R_HEAD_DW_FP16X2): the two-way fp16 form.  Unlike d X, an output row here (one item) sums the gradient entries of ONE
// column: for a rare item all of them lie 1e-6 .. 1e-12 below the tensor's maximum, so every item gets its OWN scale from a
// bound on its column (the forward kernel leaves the column maxima of the logits; DwAux).
struct DwAux { const float* colmax; const float* lse_min; const unsigned char* islab; int vpad, rsplit; };
__device__ __forceinline__ float pow2_scale_head(float m) {
    if (!(m > 0.f) || !(m < 3e38f)) return 1.f;
    int e;
    (void)frexpf(m, &e);
    return ldexpf(1.f, min(14 - e, 100));    // a maximum below 2^-86 (the d W bound of an item far below every row's lse) must not overflow the scale
}
// PD: how many 32-row blocks of the lane's logits column are in flight ahead of the one being multiplied (round 6).  The
// logits are a read-once 1.1 GB stream fetched as 4-byte column elements (a wave instruction moves two 128-byte row
// segments); with one block ahead a CU had 8 waves x 16 x 256 B = 32 KB in flight -- by Little's law ~4 TB/s at the loaded
// HBM latency, and the launch measured 3.0.  The kernel runs two waves per SIMD (two workgroups per CU), so the register
// file has room for more: PD blocks of 16 values per lane.  NT: the logits with non-temporal loads (read once: they should not
// displace the X images, which every workgroup re-reads, from L2 / Infinity Cache).  Same values in the same MFMA slots in
// the same order: d W is bit-identical for every PD / NT.
template <int NB, bool HS = false, int PD = 1, bool NT = false, bool COPY = true, int WPS = 2>
__global__ __launch_bounds__(256, WPS) void head_dw_split_kernel(const float* __restrict__ logits, long ld,
                                                             const float* __restrict__ lse, const long* __restrict__ labels,
                                                             const float* __restrict__ gout, const u32x4* __restrict__ XT,
                                                             float* __restrict__ dW, long lddw, int N, int Vc, int V,
                                                             int yoff, float smooth, float alpha, int accumulate, int nblk,
                                                             const unsigned* __restrict__ amax = nullptr, DwAux dw = DwAux()) {
    constexpr int D = 32 * NB, NPL = HS ? 2 : 3;
    constexpr int BLK = 4 * NPL * 32 * NB;
    constexpr int SN = (BLK + 255) / 256;
    __shared__ u32x4 lds[2][BLK];
    __shared__ float2 rinfo[2][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, khalf = lane >> 5;
    const int v = blockIdx.x * 128 + 32 * wave + l32;
    const float* lp = logits + min(v, Vc - 1);
    SgScalars q;
    q.g = (gout ? *gout : 1.f) / N;
    __shared__ float sh_inv[4][32];      // HS: inverse scale of every item row of the tile
    if (HS) {
        // THIS item's gradient entries are bounded by |g| (exp(max_r z[r][v] - min_r lse[r]) + eps / V), or by |g| if the item
        // is some row's label: its column gets its own power-of-two position -- rare items (entries 1e-6 .. 1e-12 of the
        // tensor's maximum) keep all 22 bits, which one scale for the whole tensor cannot give them
        // (tools/head_dw_rows_probe.py: 1e-4 relative row error there with a tensor scale, 1e-6 with item scales)
        float zmax = -INFINITY;
        for (int sidx = 0; sidx < dw.rsplit; ++sidx) zmax = fmaxf(zmax, dw.colmax[(long)sidx * dw.vpad + min(v, Vc - 1)]);
        const float pb = dw.islab[min(v, Vc - 1)] ? 1.f : __expf(fminf(zmax - *dw.lse_min, 0.f)) + smooth / V;
        const float bound = fabsf(q.g) * pb;
        const float sv = pow2_scale_head(bound);
        if (khalf == 0) sh_inv[wave][l32] = 1.f / sv;
        q.g *= sv;
        alpha = alpha / scale_of(amax);
    }
    q.sub = q.g * smooth / V;
    q.hit = q.g * (1.f - smooth);

    u32x4 st[SN];
    float ri_lse = 0.f;
    long ri_lab = 0;
    float xq[PD][16];
    // every load is unconditional and nothing computed from it appears before s_store: a guarded load (tid < 32) or a
    // conversion right behind it puts an s_waitcnt vmcnt(0) into the middle of the prefetch.  Loads return in order: the X
    // images / row facts of block b + 1 (consumed at the end of this iteration) are requested BEFORE the logits of block
    // b + PD (consumed PD iterations later), so that waiting for the former leaves the latter in flight.
    auto g_load = [&](int b) __attribute__((always_inline)) {
        const u32x4* src = XT + (long)b * BLK;
#pragma unroll
        for (int i = 0; i < SN; ++i)
            if (BLK % 256 == 0 || i * 256 + tid < BLK) st[i] = src[i * 256 + tid];
        const int row = min(b * 32 + (tid & 31), N - 1);
        ri_lse = lse[row];
        ri_lab = labels[row];
    };
    // Addresses: a block that lies wholly inside the N rows (all but the last one) is AFFINE in the row -- one 64-bit product per
    // lane and block for its first row, the other fifteen rows at compile-time multiples of the (uniform) pitch.  The clamped form
    // min(row, N - 1) * ld per element, needed only by the last block, was 48 quarter-rate integer multiplies + ~50 other address
    // instructions per lane and block: more issue cycles than the block's 24 matrix instructions (round 6, from the ISA).
    auto x_load = [&](float (&x)[16], int b) __attribute__((always_inline)) {
        if (b * 32 + 32 <= N) {             // workgroup-uniform
            const float* base = lp + (long)(b * 32 + 8 * khalf) * ld;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float* a = base + (long)(16 * s + e) * ld;
                    x[8 * s + e] = NT ? __builtin_nontemporal_load(a) : *a;
                }
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float* a = lp + (long)min(b * 32 + 16 * s + 8 * khalf + e, N - 1) * ld;
                    x[8 * s + e] = NT ? __builtin_nontemporal_load(a) : *a;
                }
        }
    };
    auto s_store = [&](int buf, int b) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < SN; ++i)
            if (BLK % 256 == 0 || i * 256 + tid < BLK) lds[buf][i * 256 + tid] = st[i];
        // rows past the end: X is zero there; exp2(-huge) = 0 keeps their gradient finite
        const bool live = b * 32 + (tid & 31) < N;
        if (tid < 32)
            rinfo[buf][tid] = live ? make_float2(ri_lse * kLog2e, __int_as_float((int)(ri_lab - yoff))) : make_float2(1e30f, __int_as_float(-1));
    };
    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    g_load(0);
#pragma unroll
    for (int u = 0; u < PD; ++u) x_load(xq[u], min(u, nblk - 1));
    s_store(0, 0);
    __syncthreads();
    for (int b0 = 0; b0 < nblk; b0 += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int b = b0 + u;
            if (b >= nblk) break;            // workgroup-uniform
            const int buf = b & 1;
            float xc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) xc[i] = xq[u][i];
            g_load(min(b + 1, nblk - 1));
            if (COPY) x_load(xq[u], min(b + PD, nblk - 1));
            __builtin_amdgcn_sched_barrier(0);
            u32x4 af[2][NPL];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float gv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float2 in = rinfo[buf][16 * s + 8 * khalf + e];
                    gv[e] = sg_value(xc[8 * s + e], in.x, __float_as_int(in.y) == v, q);
                }
                split8s<HS>(gv, 1.f, af[s]);
            }
            if (!COPY) {          // this slot's values are consumed: refill it before the products (no register copy)
                __builtin_amdgcn_sched_barrier(0);
                x_load(xq[u], min(b + PD, nblk - 1));
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    u32x4 bf[NPL];
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) bf[pl] = lds[buf][(pl * 4 + 2 * s + khalf) * D + 32 * j + l32];
                    acc[j] = mfma_split<HS>(af[s], bf, acc[j]);
                }
            s_store(buf ^ 1, min(b + 1, nblk - 1));
            __syncthreads();
        }
    }
    const int v0 = blockIdx.x * 128 + 32 * wave + 4 * khalf;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        float* cp = dW + (long)v0 * lddw + 32 * j + l32;
        float old[16];
        if (accumulate) {           // all sixteen reads in flight together
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = cp[(long)min((r & 3) + 8 * (r >> 2), Vc - 1 - v0) * lddw];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            const float a = HS ? alpha * sh_inv[wave][dr + 4 * khalf] : alpha;
            if (v0 + dr < Vc) cp[(long)dr * lddw] = a * acc[j][r] + (accumulate ? old[r] : 0.f);
        }
    }
}

// ---- d X partial sums:  part[split][N, D] = alpha * G[:, k-range] @ W[k-range, :].  grid (ceil(N / 128), splits);
// wave w owns rows 32 w .. 32 w + 31 of the tile, each lane walks its own logits row.
// HS: the two-way fp16 form (see mfma_split).  The gradient rows are scaled by a power of two taken from g / N itself; their
// tail entries (p ~ 1e-6 and below) lose relative precision in the fp16 pieces, but every OUTPUT row sums all V of them
// against well-scaled table rows: their absolute error (<= 2^-39 of the row's largest entry each) is far below the
// rounding of the sum -- unlike d W, whose rare-item rows consist of such entries only and stay on the bf16 planes.
template <int NB, bool HS = false>
__global__ __launch_bounds__(256) void head_dx_split_kernel(const float* __restrict__ logits, long ld,
                                                             const float* __restrict__ lse, const long* __restrict__ labels,
                                                             const float* __restrict__ gout, const u32x4* __restrict__ WT,
                                                             float* __restrict__ part, int N, int Vc, int V, int yoff,
                                                             float smooth, float alpha, int nkt, int kt_per, int row_tiles,
                                                             const unsigned* __restrict__ amax = nullptr) {
    constexpr int D = 32 * NB, NPL = HS ? 2 : 3;
    constexpr int BLK = 4 * NPL * 32 * NB;
    constexpr int SN = (BLK + 255) / 256;
    __shared__ u32x4 lds[2][BLK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, khalf = lane >> 5;
    // XCD-aware order: the row tiles that share one vocabulary split (one slice of the W^T planes) run on one XCD
    // (row-major order: FETCH_SIZE 1.85 GB per launch = the logits + the 77 MB of planes once per XCD)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int rt = slot % row_tiles, split = (slot / row_tiles) * 8 + xcd;
    const int kt_begin = split * kt_per, kt_end = min(nkt, kt_begin + kt_per);
    if (kt_begin >= kt_end) return;
    const int row = rt * 128 + 32 * wave + l32, rc = min(row, N - 1);
    const float* lp = logits + (long)rc * ld;
    const float l2 = lse[rc] * kLog2e;
    const int y = (int)(labels[rc] - yoff);
    SgScalars q;
    q.g = (gout ? *gout : 1.f) / N;
    if (HS) {       // |G| <= |g| (1 + eps): position g at 2^13 .. 2^14, undo it (and the table's scale) in alpha
        int e;
        (void)frexpf(fabsf(q.g), &e);
        const float sg = (q.g != 0.f && fabsf(q.g) < 3e38f) ? ldexpf(1.f, min(14 - e, 100)) : 1.f;
        q.g *= sg;
        alpha = (alpha / sg) / scale_of(amax);
    }
    q.sub = q.g * smooth / V;
    q.hit = q.g * (1.f - smooth);
    const int ldm4 = (int)ld - 4;

    u32x4 st[SN];
    float4 xn[4];
    auto g_load = [&](int kt) __attribute__((always_inline)) {
        const u32x4* src = WT + (long)kt * BLK;
#pragma unroll
        for (int i = 0; i < SN; ++i)
            if (BLK % 256 == 0 || i * 256 + tid < BLK) st[i] = src[i * 256 + tid];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int c = kt * 32 + 16 * s + 8 * khalf;       // the pitch is a multiple of 4: both loads stay inside the row
            xn[2 * s] = *reinterpret_cast<const float4*>(lp + min(c, ldm4));
            xn[2 * s + 1] = *reinterpret_cast<const float4*>(lp + min(c + 4, ldm4));
        }
    };
    auto s_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < SN; ++i)
            if (BLK % 256 == 0 || i * 256 + tid < BLK) lds[buf][i * 256 + tid] = st[i];
    };
    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    g_load(kt_begin);
    s_store(0);
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int buf = (kt - kt_begin) & 1;
        float xc[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) { xc[4 * i] = xn[i].x; xc[4 * i + 1] = xn[i].y; xc[4 * i + 2] = xn[i].z; xc[4 * i + 3] = xn[i].w; }
        g_load(min(kt + 1, kt_end - 1));
        __builtin_amdgcn_sched_barrier(0);
        if (kt * 32 + 32 > Vc) {        // vocabulary tail: columns past the end carry no gradient (their W rows are zero, keep G finite)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (kt * 32 + 16 * (i >> 3) + 8 * khalf + (i & 7) >= Vc) xc[i] = -INFINITY;
        }
        u32x4 af[2][NPL];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int dy = y - (kt * 32 + 16 * s + 8 * khalf);
            float gv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) gv[e] = sg_value(xc[8 * s + e], l2, dy == e, q);
            split8s<HS>(gv, 1.f, af[s]);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                u32x4 bf[NPL];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) bf[pl] = lds[buf][(pl * 4 + 2 * s + khalf) * D + 32 * j + l32];
                acc[j] = mfma_split<HS>(af[s], bf, acc[j]);
            }
        s_store(buf ^ 1);
        __syncthreads();
    }
    float* pp = part + (long)split * N * D;
    const int r0 = rt * 128 + 32 * wave + 4 * khalf;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = r0 + (r & 3) + 8 * (r >> 2);
            if (rr < N) pp[(long)rr * D + 32 * j + l32] = alpha * acc[j][r];
        }
}

// =====================================================================================================================
// The RECOMPUTING backward (round 4; two-way fp16 form only).  The materialised head writes 1.1 GB of logits and reads them
// twice (d X, d W): 3.6 GB of HBM traffic per step for 2.3 GB of need, and all three kernels run at "the logits' HBM time
// plus their matrix time plus their VALU time" -- the parts do not overlap on this chip.  Here nothing of size [N, V] ever
// exists: the forward keeps statistics only (head_ce_stats_h_kernel), and each backward kernel RECOMPUTES its score tile on
// the matrix cores -- three more matrix instructions per K = 16 step instead of a 128-byte row segment from HBM -- in the
// orientation whose ACCUMULATOR layout is, after the softmax-gradient transform in registers, the A operand of the product
// that follows (the contraction index in accumulator order: the KMP image above).  Arithmetic: the same products, the same
// exp / cut per gradient element as the materialised kernels; the recomputed logits repeat the forward's (same operand
// pieces, same order of the three partial products and of K).
//
// d W: one workgroup per 128 items, all label rows (as head_dw_split_kernel): W rows as B fragments in registers (cut once),
//      X blocks through LDS twice -- MK image (A operand of the scores) and KMP image (B operand of d W).
template <int NB>
__global__ __launch_bounds__(256) void head_dw_rc_kernel(const u32x4* __restrict__ XA, const u32x4* __restrict__ XTP,
                                                          const float* __restrict__ W, long ldw, const float* __restrict__ lse,
                                                          const long* __restrict__ labels, const float* __restrict__ gout,
                                                          float* __restrict__ dW, long lddw, int N, int V, float smooth, float alpha,
                                                          int accumulate, int nblk, const unsigned* __restrict__ amax, DwAux dw) {
    constexpr int D = 32 * NB, KS = 2 * NB, CH = 4 * NB;
    constexpr int BLK = 4 * 2 * 32 * NB;            // u32x4 per two-plane block (either image)
    constexpr int SN = (BLK + 255) / 256;
    __shared__ u32x4 ldsA[2][BLK];
    __shared__ u32x4 ldsT[2][BLK];
    __shared__ float2 rinfo[2][32];
    __shared__ float sh_inv[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, khalf = lane >> 5;
    const int v = blockIdx.x * 128 + 32 * wave + l32, vc = min(v, V - 1);
    const float sx = scale_of(amax), sw = scale_of(amax + 1);
    const float zscale = (alpha / sx) / sw;          // accumulator -> logit
    u32x4 Bf[KS][2];
    {
        const float* wr = W + (long)vc * ldw + 8 * khalf;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 u = *reinterpret_cast<const float4*>(wr + 16 * s);
            const float4 t = *reinterpret_cast<const float4*>(wr + 16 * s + 4);
            const float x[8] = {u.x, u.y, u.z, u.w, t.x, t.y, t.z, t.w};
            split8s<true>(x, sw, Bf[s]);
        }
    }
    SgScalars q;
    q.g = (gout ? *gout : 1.f) / N;
    {   // this item's power-of-two position from a bound on its column (head_dw_split_kernel)
        float zmax = -INFINITY;
        for (int sidx = 0; sidx < dw.rsplit; ++sidx) zmax = fmaxf(zmax, dw.colmax[(long)sidx * dw.vpad + vc]);
        const float pb = dw.islab[vc] ? 1.f : __expf(fminf(zmax - *dw.lse_min, 0.f)) + smooth / V;
        const float sv = pow2_scale_head(fabsf(q.g) * pb);
        if (khalf == 0) sh_inv[wave][l32] = 1.f / sv;
        q.g *= sv;
    }
    q.sub = q.g * smooth / V;
    q.hit = q.g * (1.f - smooth);
    const float oscale = alpha / sx;                 // d W = alpha G^T X: undo X's scale (the item scale per output row)

    u32x4 stA[SN], stT[SN];
    float ri_lse = 0.f;
    long ri_lab = 0;
    auto g_load = [&](int b) __attribute__((always_inline)) {
        const int row = min(b * 32 + (tid & 31), N - 1);
        ri_lse = lse[row];
        ri_lab = labels[row];
        const u32x4* sa = XA + (long)b * BLK;
        const u32x4* stp = XTP + (long)b * BLK;
#pragma unroll
        for (int i = 0; i < SN; ++i)
            if (BLK % 256 == 0 || i * 256 + tid < BLK) { stA[i] = sa[i * 256 + tid]; stT[i] = stp[i * 256 + tid]; }
    };
    auto s_store = [&](int buf, int b) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < SN; ++i)
            if (BLK % 256 == 0 || i * 256 + tid < BLK) { ldsA[buf][i * 256 + tid] = stA[i]; ldsT[buf][i * 256 + tid] = stT[i]; }
        const bool live = b * 32 + (tid & 31) < N;
        if (tid < 32)
            rinfo[buf][tid] = live ? make_float2(ri_lse * kLog2e, __int_as_float((int)ri_lab)) : make_float2(1e30f, __int_as_float(-1));
    };
    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    g_load(0);
    s_store(0, 0);
    __syncthreads();
    for (int b = 0; b < nblk; ++b) {
        const int buf = b & 1;
        g_load(min(b + 1, nblk - 1));
        __builtin_amdgcn_sched_barrier(0);
        // the score tile: rows of the block x this wave's 32 items; lane (item, khalf) gets rows (r & 3) + 8 (r >> 2) + 4 khalf
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 a[2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) a[pl] = ldsA[buf][(pl * CH + 2 * s + khalf) * 32 + l32];
            z = mfma_split_swapped(a, Bf[s], z);
        }
        u32x4 af[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float gv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 8 * s + e;
                const float2 in = rinfo[buf][(r & 3) + 8 * (r >> 2) + 4 * khalf];
                gv[e] = sg_value(zscale * z[r], in.x, __float_as_int(in.y) == v, q);
            }
            split8s<true>(gv, 1.f, af[s]);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                u32x4 bf[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) bf[pl] = ldsT[buf][(pl * 4 + 2 * s + khalf) * D + 32 * j + l32];
                acc[j] = mfma_split<true>(af[s], bf, acc[j]);
            }
        s_store(buf ^ 1, min(b + 1, nblk - 1));
        __syncthreads();
    }
    const int v0 = blockIdx.x * 128 + 32 * wave + 4 * khalf;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        float* cp = dW + (long)v0 * lddw + 32 * j + l32;
        float old[16];
        if (accumulate) {
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = cp[(long)min((r & 3) + 8 * (r >> 2), V - 1 - v0) * lddw];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            const float a = oscale * sh_inv[wave][dr + 4 * khalf];
            if (v0 + dr < V) cp[(long)dr * lddw] = a * acc[j][r] + (accumulate ? old[r] : 0.f);
        }
    }
}

// d X: workgroups of 128 label rows x a range of 32-item tiles (as head_dx_split_kernel): the rows of X as B fragments in
//      registers (cut once), the table's tiles through LDS twice -- MK image (A operand of the scores: lane = item) and KMP
//      image (B operand of d X); lane (row, khalf) gets the items (r & 3) + 8 (r >> 2) + 4 khalf of the tile.
template <int NB>
__global__ __launch_bounds__(256) void head_dx_rc_kernel(const float* __restrict__ X, long ldx, const u32x4* __restrict__ WA,
                                                          const u32x4* __restrict__ WTP, const float* __restrict__ lse,
                                                          const long* __restrict__ labels, const float* __restrict__ gout,
                                                          float* __restrict__ part, int N, int V, float smooth, float alpha,
                                                          int nkt, int kt_per, int row_tiles, const unsigned* __restrict__ amax) {
    constexpr int D = 32 * NB, KS = 2 * NB, CH = 4 * NB;
    constexpr int BLK = 4 * 2 * 32 * NB;
    constexpr int SN = (BLK + 255) / 256;
    __shared__ u32x4 ldsA[2][BLK];
    __shared__ u32x4 ldsT[2][BLK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, khalf = lane >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int rt = slot % row_tiles, split = (slot / row_tiles) * 8 + xcd;
    const int kt_begin = split * kt_per, kt_end = min(nkt, kt_begin + kt_per);
    if (kt_begin >= kt_end) return;
    const int row = rt * 128 + 32 * wave + l32, rc = min(row, N - 1);
    const float sx = scale_of(amax), sw = scale_of(amax + 1);
    const float zscale = (alpha / sx) / sw;
    u32x4 Xf[KS][2];
    {
        const float* xr = X + (long)rc * ldx + 8 * khalf;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 u = *reinterpret_cast<const float4*>(xr + 16 * s);
            const float4 t = *reinterpret_cast<const float4*>(xr + 16 * s + 4);
            const float x[8] = {u.x, u.y, u.z, u.w, t.x, t.y, t.z, t.w};
            split8s<true>(x, sx, Xf[s]);
        }
    }
    const float l2 = lse[rc] * kLog2e;
    const int y = (int)labels[rc];
    SgScalars q;
    q.g = (gout ? *gout : 1.f) / N;
    float oscale;
    {
        int e;
        (void)frexpf(fabsf(q.g), &e);
        const float sg = (q.g != 0.f && fabsf(q.g) < 3e38f) ? ldexpf(1.f, min(14 - e, 100)) : 1.f;
        q.g *= sg;
        oscale = (alpha / sg) / sw;
    }
    q.sub = q.g * smooth / V;
    q.hit = q.g * (1.f - smooth);

    u32x4 stA[SN], stT[SN];
    auto g_load = [&](int kt) __attribute__((always_inline)) {
        const u32x4* sa = WA + (long)kt * BLK;
        const u32x4* stp = WTP + (long)kt * BLK;
#pragma unroll
        for (int i = 0; i < SN; ++i)
            if (BLK % 256 == 0 || i * 256 + tid < BLK) { stA[i] = sa[i * 256 + tid]; stT[i] = stp[i * 256 + tid]; }
    };
    auto s_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < SN; ++i)
            if (BLK % 256 == 0 || i * 256 + tid < BLK) { ldsA[buf][i * 256 + tid] = stA[i]; ldsT[buf][i * 256 + tid] = stT[i]; }
    };
    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    g_load(kt_begin);
    s_store(0);
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int buf = (kt - kt_begin) & 1;
        g_load(min(kt + 1, kt_end - 1));
        __builtin_amdgcn_sched_barrier(0);
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 a[2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) a[pl] = ldsA[buf][(pl * CH + 2 * s + khalf) * 32 + l32];
            z = mfma_split<true>(a, Xf[s], z);
        }
        u32x4 af[2][2];
        const int dy = y - kt * 32 - 4 * khalf;          // the label's position among this lane's items
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float gv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = 8 * s + e, c = (r & 3) + 8 * (r >> 2);
                // items past the end of the vocabulary carry no gradient (their table rows are zero in both images)
                const float zz = (kt * 32 + c + 4 * khalf < V) ? zscale * z[r] : -INFINITY;
                gv[e] = sg_value(zz, l2, dy == c, q);
            }
            split8s<true>(gv, 1.f, af[s]);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                u32x4 bf[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) bf[pl] = ldsT[buf][(pl * 4 + 2 * s + khalf) * D + 32 * j + l32];
                acc[j] = mfma_split<true>(af[s], bf, acc[j]);
            }
        s_store(buf ^ 1);
        __syncthreads();
    }
    float* pp = part + (long)split * N * D;
    const int r0 = rt * 128 + 32 * wave + 4 * khalf;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = r0 + (r & 3) + 8 * (r >> 2);
            if (rr < N) pp[(long)rr * D + 32 * j + l32] = oscale * acc[j][r];
        }
}

// out[row, :] (+)= sum over the splits, in split order (deterministic)
__global__ __launch_bounds__(256) void head_dx_reduce_kernel(const float* __restrict__ part, int n_split, long nd4, int d4,
                                                              float* __restrict__ out, long ldo, int accumulate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nd4) return;
    float4 s = reinterpret_cast<const float4*>(part)[i];
#pragma unroll 8
    for (int k = 1; k < n_split; ++k) {      // eight requests in flight, added in split order
        const float4 t = reinterpret_cast<const float4*>(part)[(long)k * nd4 + i];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    float4* op = reinterpret_cast<float4*>(out + (i / d4) * ldo + (i % d4) * 4);
    if (accumulate) { const float4 o = *op; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    *op = s;
}

// =====================================================================================================================
// Logits + cross-entropy statistics + d X in ONE pass (round 5; two-way fp16 form).  The materialised head wrote the 1.1 GB of
// logits once and read them twice (d X, d W): 3.65 GB of HBM traffic per step for 2.3 GB of need, three rounds running
// (VERDICT r4 weak #5).  The round-4 argument that a one-pass BACKWARD is infeasible (partial sums the size of the logits) holds
// for item-block- and row-tile-stationary backward kernels; what it leaves open is forming d X in the FORWARD, flash-attention
// style: a workgroup owns 128 label rows (X as B fragments in registers) and a RANGE of 32-item tiles of the table, and per tile
//   * recomputes nothing: the score tile S = W_tile X^T comes off the matrix cores once (A = MK image of the table tile:
//     lane = item; accumulator lane (row, khalf) = sixteen items of ONE row), is scaled, stored as logits (full 128-byte row
//     segments through the quad transpose of head_logits_ce_body) and reduced to the column maxima d W's per-item scales need;
//   * keeps the row's running softmax reference m_ref (bumped only when a score exceeds it by more than 2^5: a wave-uniform,
//     rare branch that rescales the accumulators) and sum s = sum exp(z - m_ref);
//   * feeds P~ = exp(z - m_ref) <= 32, positioned at 2^9 and cut into two fp16 pieces IN REGISTERS, to the second product
//     as its B operand (n = row) against the KMP image of the tile as A (m = feature): the d X accumulator of lane (row, khalf)
//     holds features of its OWN row, so the reference bump is a lane-local multiply.
// Per (row tile, item range) it leaves (m_ref, s, sum of logits) per row and a [128, D] partial of sum_v P~_v W_v: 23 splits x
// 1.4 MB at BASELINE configs[1] instead of a second 1.1 GB read.  head_fdx_finalize_kernel merges the statistics into lse and
// the loss and turns the partials into d X for grad_out = 1:
//   d X[r] = alpha / N * ( sum_c e^{m_c - lse_r} part_c[r] / (2^9 s_W) - (1 - eps) W[y_r] - eps / V * colsum(W) ),
// the label row and the smoothing term in exact fp32, outside the matrix product.  d W stays head_dw_split_kernel (one read of
// the logits).  Replaces, with autograd's d X of it, transformers4rec/torch/model/prediction_task.py:648-671 + CE :446.
constexpr float kFdxTau = 3.4657359f;        // 5 ln 2: the reference is bumped when a score exceeds it by more than 2^5
constexpr float kFdxPScale = 512.f;          // P~ <= 2^5 -> P~ 2^9 <= 2^14 (the fp16 position of mfma_split)

__device__ __forceinline__ float dpp_row_shr_max(float x, int ctrl4) {
    // max(x, x of the lane `n` positions lower in its row of 16); lanes without a source keep x
    const int y = ctrl4 == 4 ? __builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x114, 0xf, 0xf, false)
                             : __builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x118, 0xf, 0xf, false);
    return fmaxf(x, __int_as_float(y));
}

// launch_bounds(256, 2): TWO waves per SIMD.  Left alone the compiler took 213 registers + 80 accumulation registers = ONE wave
// per SIMD, i.e. one workgroup per CU although the 66 KB of LDS let two in (the host code below launches one residency of
// two per CU: the 506 workgroups of BASELINE configs[1] ran as two rounds); asked for two waves it fits 222 registers with no
// spill, both workgroups are resident and each fills the other's waits: 723 -> 609 us stand-alone, 2.83 -> 2.70 ms per step
// (profiles/r05_y_ab.txt).  The same request on head_dw_split_kernel (3 waves, 161 registers) measured 0.02 ms slower: not kept.
// SMOOTH: label smoothing on (the row sums of the logits are wanted: st_t non-null).  Without it the sixteen adds + selects per
// tile and lane that form those sums are not compiled in (round 6: the kernel's time is matrix + VECTOR issue).
template <int NB, bool SMOOTH>
__global__ __launch_bounds__(256, 2) void head_fwd_dx_kernel(const float* __restrict__ X, long ldx, const u32x4* __restrict__ WA,
                                                          const u32x4* __restrict__ WTP, float* __restrict__ C, long ldc,
                                                          int vec_ok, float* __restrict__ part, float* __restrict__ st_m,
                                                          float* __restrict__ st_s, float* __restrict__ st_t,
                                                          float* __restrict__ colmax, int vpad, int N, int V, float alpha,
                                                          int nkt, int kt_per, int row_tiles, int n_wg, int per_xcd,
                                                          const unsigned* __restrict__ amax) {
    constexpr int D = 32 * NB, KS = 2 * NB, CH = 4 * NB;
    constexpr int BLK = 4 * 2 * 32 * NB;
    constexpr int SN = (BLK + 255) / 256;
    __shared__ u32x4 ldsA[2][BLK];
    __shared__ u32x4 ldsT[2][BLK];
    __shared__ float4 sh_cm[2][4][2][8];        // [buffer][wave][16-row half][slot q: items 4 q .. 4 q + 3 of the tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, khalf = lane >> 5;
    // Workgroup order: linear id = split * row_tiles + row tile (the row tiles of one item range adjacent), and XCD x (the
    // workgroups with blockIdx % 8 == x) takes the ids [x per_xcd, (x + 1) per_xcd): every XCD gets the SAME number of
    // workgroups -- at most its 64 resident slots at BASELINE configs[1] -- and streams the table images of at most
    // ceil(per_xcd / row_tiles) + 1 item ranges through its L2.  (The per-split XCD assignment of head_dx_split_kernel gave
    // 23 splits as 3, 3, .., 2 per XCD = 66 workgroups on seven XCDs for 64 slots: a second round for two of them, 970 us
    // instead of 490.)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int lin = xcd * per_xcd + slot;
    if (slot >= per_xcd || lin >= n_wg) return;
    const int split = lin / row_tiles, rt = lin - split * row_tiles;
    const int kt_begin = split * kt_per, kt_end = min(nkt, kt_begin + kt_per);
    if (kt_begin >= kt_end) return;
    const int row = rt * 128 + 32 * wave + l32, rc = min(row, N - 1);
    const float sx = scale_of(amax), sw = scale_of(amax + 1);
    const float zscale = (alpha / sx) / sw;          // accumulator -> logit (exact powers of two)
    u32x4 Xf[KS][2];
    {
        const float* xr = X + (long)rc * ldx + 8 * khalf;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 u = *reinterpret_cast<const float4*>(xr + 16 * s);
            const float4 t = *reinterpret_cast<const float4*>(xr + 16 * s + 4);
            const float x[8] = {u.x, u.y, u.z, u.w, t.x, t.y, t.z, t.w};
            split8s<true>(x, sx, Xf[s]);
        }
    }
    u32x4 stA[SN], stT[SN];
    auto g_load = [&](int kt) __attribute__((always_inline)) {
        const u32x4* sa = WA + (long)kt * BLK;
        const u32x4* stp = WTP + (long)kt * BLK;
#pragma unroll
        for (int i = 0; i < SN; ++i)
            if (BLK % 256 == 0 || i * 256 + tid < BLK) { stA[i] = sa[i * 256 + tid]; stT[i] = stp[i * 256 + tid]; }
    };
    auto s_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < SN; ++i)
            if (BLK % 256 == 0 || i * 256 + tid < BLK) { ldsA[buf][i * 256 + tid] = stA[i]; ldsT[buf][i * 256 + tid] = stT[i]; }
    };
    // column maxima of one tile: the eight partials per slot (4 waves x 2 row halves) -> one float4 per slot (threads 0..7)
    auto flush_cm = [&](int buf, int kt) __attribute__((always_inline)) {
        if (tid < 8) {
            float4 a = sh_cm[buf][0][0][tid];
#pragma unroll
            for (int k = 1; k < 8; ++k) {
                const float4 o = sh_cm[buf][k >> 1][k & 1][tid];
                a.x = fmaxf(a.x, o.x); a.y = fmaxf(a.y, o.y); a.z = fmaxf(a.z, o.z); a.w = fmaxf(a.w, o.w);
            }
            *reinterpret_cast<float4*>(colmax + (long)rt * vpad + kt * 32 + 4 * tid) = a;
        }
    };
    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float m_ref = -INFINITY, s_run = 0.f, t_run = 0.f;
    const int t4 = l32 & 3;
    const int rq = rt * 128 + 32 * wave + (l32 & ~3);           // first row of this lane's quad
    g_load(kt_begin);
    s_store(0);
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int buf = (kt - kt_begin) & 1;
        g_load(min(kt + 1, kt_end - 1));
        __builtin_amdgcn_sched_barrier(0);
        if (colmax && kt > kt_begin) flush_cm(buf ^ 1, kt - 1);
        // score tile: this wave's 32 rows x the tile's 32 items; lane (row, khalf) gets items (r & 3) + 8 (r >> 2) + 4 khalf
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 a[2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) a[pl] = ldsA[buf][(pl * CH + 2 * s + khalf) * 32 + l32];
            z = mfma_split<true>(a, Xf[s], z);
        }
        // the next tile into LDS before the logits are stored (its loads were issued at the top): the wait for them must not
        // cover the HBM stores below, and the staging registers die here.  (Measured, round 5: moving this to the END of the
        // step -- a whole step of latency cover, made exact by unconditional clamped-row stores and a peeled vocabulary tail:
        // s_waitcnt vmcnt(4) in the ISA -- costs 39 more VGPRs (252) and is 0.5 % SLOWER per step, 712 vs 700 us on one box:
        // the kernel is not waiting on these loads; its time is matrix + vector + LDS issue, which add up on this chip.)
        s_store(buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        float v[16];
        const int c0 = kt * 32 + 4 * khalf;                     // this lane's items: c0 + 8 g + i, r = 4 g + i
        const bool tail = kt * 32 + 32 > V;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = zscale * z[r];
        if (tail) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (c0 + 8 * (r >> 2) + (r & 3) >= V) v[r] = -INFINITY;       // past the vocabulary: no score, no probability
        }
        // rows of the quad x item groups through DPP: w[g][i] = (row rq + g, item kt 32 + 8 t4 + 4 khalf + i)
        float w[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float x[4] = {v[i], v[4 + i], v[8 + i], v[12 + i]};
            quad_transpose4(x, t4);
#pragma unroll
            for (int g = 0; g < 4; ++g) w[g][i] = x[g];
        }
        if (vec_ok && !tail) {
            float* cq = C + (long)rq * ldc + kt * 32 + 8 * t4 + 4 * khalf;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (rq + g < N) {
                    const f32x4 o = {w[g][0], w[g][1], w[g][2], w[g][3]};
                    __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(cq + (long)g * ldc));
                }
        } else if (row < N) {
            float* cr = C + (long)row * ldc + c0;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (c0 + 8 * (r >> 2) + (r & 3) < V) cr[8 * (r >> 2) + (r & 3)] = v[r];
        }
        if (colmax) {
            // (rows past N repeat row N - 1: they cannot raise a maximum)
            float cm[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                cm[i] = fmaxf(fmaxf(w[0][i], w[1][i]), fmaxf(w[2][i], w[3][i]));
                cm[i] = dpp_row_shr_max(cm[i], 4);
                cm[i] = dpp_row_shr_max(cm[i], 8);          // lanes 12 .. 15 of every row of 16: the four quads merged
            }
            if ((l32 & 12) == 12) sh_cm[buf][wave][l32 >> 4][2 * t4 + khalf] = make_float4(cm[0], cm[1], cm[2], cm[3]);
        }
        // running reference of the row (both lanes of a row share it) and the probabilities against it
        float mloc = v[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, v[r]);
        const float mrow = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        if (__builtin_amdgcn_ballot_w64(mrow > m_ref + kFdxTau) != 0) {        // wave-uniform, rare after the first tiles
            const float nref = mrow > m_ref + kFdxTau ? mrow : m_ref;
            const float a = nref == m_ref ? 1.f : __builtin_amdgcn_exp2f((m_ref - nref) * kLog2e);
            s_run *= a;
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] *= a;
            m_ref = nref;
        }
        const float ref2 = m_ref * kLog2e;
        u32x4 af[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float pv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = v[8 * s + e];
                pv[e] = __builtin_amdgcn_exp2f(fmaf(x, kLog2e, -ref2));
                s_run += pv[e];
                if (SMOOTH) t_run += (tail && x == -INFINITY) ? 0.f : x;
            }
            split8s<true>(pv, kFdxPScale, af[s]);
        }
        // d X^T (features x rows) += W_tile^T (A: KMP image) . P~ (B: this lane's row)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                u32x4 bf[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) bf[pl] = ldsT[buf][(pl * 4 + 2 * s + khalf) * D + 32 * j + l32];
                acc[j] = mfma_split<true>(bf, af[s], acc[j]);
            }
        __syncthreads();
    }
    if (colmax) flush_cm((kt_end - 1 - kt_begin) & 1, kt_end - 1);
    // partial d X: lane (row, khalf) holds features 32 j + 8 g + 4 khalf + (0..3) of its row
    if (row < N) {
        float* pp = part + ((long)split * N + row) * D + 4 * khalf;
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(pp + 32 * j + 8 * g) = make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
    }
    const float s_tot = s_run + __shfl_xor(s_run, 32, 64), t_tot = t_run + __shfl_xor(t_run, 32, 64);
    if (khalf == 0 && row < N) {
        const long o = (long)split * N + row;
        st_m[o] = m_ref;
        st_s[o] = s_tot;
        if (SMOOTH) st_t[o] = t_tot;
    }
}

// statistics of the item ranges -> lse, loss per row; partials -> d X for grad_out = 1 (see above).  256 threads = 8 rows x 32
// feature quads; the splits are walked in order (deterministic).
__global__ __launch_bounds__(256) void head_fdx_finalize_kernel(const float* __restrict__ st_m, const float* __restrict__ st_s,
                                                                 const float* __restrict__ st_t, const float* __restrict__ part,
                                                                 int n_split, int N, int V, int D, const float* __restrict__ C,
                                                                 long ldc, const float* __restrict__ W, long ldw,
                                                                 const long* __restrict__ labels, const float* __restrict__ wsum,
                                                                 float smoothing, float alpha, const unsigned* __restrict__ amax,
                                                                 float* __restrict__ loss_rows, float* __restrict__ lse_out,
                                                                 float* __restrict__ dX, long lddx) {
    __shared__ float sh_f[8][64];
    const int r = threadIdx.x >> 5, c = threadIdx.x & 31;
    const int row = blockIdx.x * 8 + r, rc = min(row, N - 1);
    float m = -INFINITY, s = 0.f, t = 0.f;
    for (int k = c; k < n_split; k += 32) {
        const long o = (long)k * N + rc;
        lse_merge(m, s, st_m[o], st_s[o]);
        if (st_t) t += st_t[o];
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        lse_merge(m, s, __shfl_xor(m, o, 64), __shfl_xor(s, o, 64));
        t += __shfl_xor(t, o, 64);
    }
    const float lse = m + logf(s);
    const long y = labels[rc];
    for (int k = c; k < n_split; k += 32) sh_f[r][k] = __expf(st_m[(long)k * N + rc] - lse);
    if (c == 0 && row < N) {
        float loss = lse - C[(long)row * ldc + y];
        if (smoothing > 0.f) loss = (1.f - smoothing) * loss + smoothing * (lse - t / V);
        loss_rows[row] = loss;
        lse_out[row] = lse;
    }
    __syncthreads();
    if (row >= N || 4 * c >= D || dX == nullptr) return;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < n_split; ++k) {
        const float4 p = *reinterpret_cast<const float4*>(part + ((long)k * N + row) * D + 4 * c);
        const float f = sh_f[r][k];
        a.x = fmaf(p.x, f, a.x); a.y = fmaf(p.y, f, a.y); a.z = fmaf(p.z, f, a.z); a.w = fmaf(p.w, f, a.w);
    }
    const float un = (1.f / kFdxPScale) / scale_of(amax + 1);       // undo the positions of P~ and of the table (powers of two)
    const float4 wy = *reinterpret_cast<const float4*>(W + y * ldw + 4 * c);
    const float hit = 1.f - smoothing, g = alpha / N;
    float4 o = make_float4(a.x * un - hit * wy.x, a.y * un - hit * wy.y, a.z * un - hit * wy.z, a.w * un - hit * wy.w);
    if (smoothing > 0.f) {
        const float4 ws4 = *reinterpret_cast<const float4*>(wsum + 4 * c);
        const float q = smoothing / V;
        o.x -= q * ws4.x; o.y -= q * ws4.y; o.z -= q * ws4.z; o.w -= q * ws4.w;
    }
    *reinterpret_cast<float4*>(dX + (long)row * lddx + 4 * c) = make_float4(g * o.x, g * o.y, g * o.z, g * o.w);
}

// both images of the table the one-pass forward streams (MK: A operand of the scores; KMP: A operand of d X), one launch
// amax_part (round 6, optional): the table's maximum arrives as <= 1024 per-workgroup partials the optimizer left behind
// (t4r_adam_step_amax) instead of in *amax: every workgroup reduces them (4 KB of L2 hits), workgroup (0, 0) stores the result
// where the later kernels of the head read it
template <int NB>
__global__ __launch_bounds__(256) void split_w_images_kernel(const float* __restrict__ src, long ld, int n_rows,
                                                              u32x4* __restrict__ wa, u32x4* __restrict__ wtp,
                                                              unsigned* amax_word, const float* __restrict__ amax_part,
                                                              int n_part) {
    constexpr int D = 32 * NB;
    __shared__ unsigned sh_bits;
    __shared__ float sh_red[4];
    const unsigned* amax = amax_word;
    if (amax_part) {                 // workgroup-uniform
        float mx = 0.f;
        for (int i = threadIdx.x; i < n_part; i += 256) mx = fmaxf(mx, amax_part[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if ((threadIdx.x & 63) == 0) sh_red[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            sh_bits = __float_as_uint(fmaxf(fmaxf(sh_red[0], sh_red[1]), fmaxf(sh_red[2], sh_red[3])));
            if (blockIdx.x == 0 && blockIdx.y == 0) *amax_word = sh_bits;
        }
        __syncthreads();
        amax = &sh_bits;
    }
    const int b = blockIdx.x;
    if (blockIdx.y == 0) { split_mk_body<NB, true>(b, src, ld, n_rows, wa, amax); return; }
    const float scale = scale_of(amax);
    for (int idx = threadIdx.x; idx < 4 * D; idx += 256) {
        const int d = idx % D, kc = idx / D;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int row = b * 32 + 16 * (kc >> 1) + 4 * (kc & 1) + (e & 3) + 8 * (e >> 2);
            x[e] = row < n_rows ? src[(long)row * ld + d] : 0.f;
        }
        u32x4 w[2];
        split8s<true>(x, scale, w);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) wtp[(((long)b * 2 + pl) * 4 + kc) * D + d] = w[pl];
    }
}

static int head_rows_per_wg() {
    static int per = -1;
    if (per < 0) { const char* e = t4r_exp_getenv("T4R_HEAD_ROWS_PER_WG"); per = e ? max(1, atoi(e)) : 12; }
    return per;
}
static int head_dx_target() {
    static int target = -1;
    if (target < 0) { const char* e = t4r_exp_getenv("T4R_HEAD_DX_WGS"); target = e ? atoi(e) : 1536; }
    return target;
}
// workspace layout (bytes): XA | XT | WT | d X partials
struct HeadWs { long xa, xt, wt, part, stats, scales, xth, colmax, islab, xtp, wa, zlab, total; int nblk, nkt, max_split, ntile, vpad, rsplit, cmrows; };
// d W too (T4R_HEAD_DW_FP16X2, default 1; per-item scales: see head_dw_split_kernel)?
static bool head_dw_fp16x2() {
    static int on = -1;
    if (on < 0) { const char* e = t4r_exp_getenv("T4R_HEAD_DW_FP16X2"); on = e ? (atoi(e) != 0) : 1; }
    return on != 0;
}
// the forward product in the two-way fp16 split (see mfma_split)?
static bool head_fwd_fp16x2() {
    static int on = -1;
    if (on < 0) { const char* e = t4r_exp_getenv("T4R_HEAD_FWD_FP16X2"); on = e ? (atoi(e) != 0) : 1; }
    return on != 0;
}
// the recomputing form available (both fp16 switches on) and not switched off (T4R_HEAD_RECOMPUTE=0 also drops its 75 MB
// of table planes from the workspace)?
static bool head_recompute_on() {
    static int on = -1;
    if (on < 0) { const char* e = t4r_exp_getenv("T4R_HEAD_RECOMPUTE"); on = e ? (atoi(e) != 0) : 1; }
    return on != 0 && head_fwd_fp16x2() && head_dw_fp16x2();
}
// logits + statistics + d X in one pass (head_fwd_dx_kernel; T4R_HEAD_FDX=0 restores logits-then-d X-from-the-logits)?
static bool head_fdx_on() {
    static int on = -1;
    if (on < 0) { const char* e = t4r_exp_getenv("T4R_HEAD_FDX"); on = e ? (atoi(e) != 0) : 1; }
    return on != 0 && head_fwd_fp16x2();
}
HeadWs head_ws(int N, int V, int D) {
    HeadWs w;
    w.nblk = (N + 31) / 32;
    w.nkt = (V + 31) / 32;
    const long blk = 12L * D * 16;
    // upper bound of the d X split count t4r_head_split_dx can choose for these sizes (the same rule: <= 64, >= 8 k-tiles
    // per split, ~T4R_HEAD_DX_WGS workgroups in total): the partial buffer is sized for it, not for 64 always
    // (64 x N x D x 4 bytes was 1.1 GB of mostly unused workspace at N = 35 k, D = 128, held from forward to backward)
    w.max_split = max(1, min(min(64, w.nkt / 8), head_dx_target() / ((N + 127) / 128)));
    w.xa = 0;
    w.xt = w.xa + w.nblk * blk;
    w.wt = w.xt + w.nblk * blk;
    w.part = w.wt + w.nkt * blk;
    w.stats = w.part + (long)w.max_split * N * D * 4;
    w.ntile = (V + 127) / 128;
    w.scales = w.stats + 3L * w.ntile * N * 4;       // words: bits of max |X|, bits of max |W| (fp16 split scales), min lse
    // d W in the fp16 form (per-item scales): the fp16 K-major planes of X, the per-row-split column maxima of the logits
    // the forward kernel leaves, a byte per item "is some row's label"
    w.xth = w.scales + 256;
    w.vpad = 128 * w.ntile;
    w.rsplit = (w.nblk + head_rows_per_wg() - 1) / head_rows_per_wg();
    w.colmax = w.xth + w.nblk * blk;
    w.cmrows = max(w.rsplit, (N + 127) / 128);          // the one-pass forward (head_fwd_dx_kernel) leaves one row per 128-row tile
    w.islab = w.colmax + (long)w.cmrows * w.vpad * 4;
    // the recomputing head (round 4): X in accumulator order (KMP), the table as MK blocks (its KMP blocks take `wt`), label logits
    w.xtp = w.islab + ((w.vpad + 255) / 256) * 256;
    w.wa = w.xtp + w.nblk * blk;
    w.zlab = w.wa + ((head_recompute_on() || head_fdx_on()) ? w.nkt * blk : 0);
    w.total = w.zlab + (((long)N * 4 + 255) / 256) * 256;
    return w;
}
bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

// What a forward product leaves for its backward products (which table slice the max |W| word of the workspace describes,
// whether the workspace holds the column maxima of THESE logits): a caller-owned host struct (include/t4r_hip.h:
// t4r_head_note, 64 bytes, zeroed by the caller before the forward), handed from _logits / _logits_ce to _dx / _dw.  The
// library keeps nothing between calls.  note == NULL: the backward products assume nothing (d X finds max |W| itself, d W
// runs on the three bf16 planes).
struct FwdNote { const float* W; const float* logits; int Vw, V, N, colmax, dw_form, cm_rows; };     // cm_rows: rows of column maxima (0: the row splits of head_logits_ce)
static_assert(sizeof(FwdNote) <= 64, "t4r_head_note is 64 bytes");
static inline FwdNote* note_of(void* p) { return reinterpret_cast<FwdNote*>(p); }
extern "C" int t4r_head_note_dw_form(const void* note) { return note ? reinterpret_cast<const FwdNote*>(note)->dw_form : 0; }
static int head_w_amax(hipStream_t st, const float* W, long ldw, int V, int D, unsigned* out) {
    if (hipMemsetAsync(out, 0, 4, st) != hipSuccess) { t4r_set_error("head_split: memset failed"); return -1; }
    const long n4 = (long)V * (D / 4);
    hipLaunchKernelGGL(amax_kernel, dim3((unsigned)min(256L, (n4 + 1023) / 1024)), dim3(1024), 0, st, W, ldw, (long)V, D, out);
    return 0;
}

#define T4R_NB_SWITCH(D, CALL)                \
    switch ((D) / 32) {                       \
        case 1: { constexpr int NB = 1; CALL; } break; \
        case 2: { constexpr int NB = 2; CALL; } break; \
        case 3: { constexpr int NB = 3; CALL; } break; \
        default: { constexpr int NB = 4; CALL; } break; \
    }

// 1 when these kernels take the shape (the callers fall back to the general GEMM otherwise)
extern "C" int t4r_head_split_supported(int D) { return D >= 32 && D <= 128 && D % 32 == 0; }
// matrix instructions per fp32-equivalent one in the forward / d X products: 3 (two-way fp16 split) or 6 (three bf16 planes)
extern "C" int t4r_head_split_fwd_products(void) { return head_fwd_fp16x2() ? 3 : 6; }

extern "C" long t4r_head_split_ws_bytes(int N, int V, int D) {
    if (!t4r_head_split_supported(D) || N <= 0 || V <= 0) return 0;
    return head_ws(N, V, D).total;
}

// cuts X [N, D] (the head's input rows) into the plane blocks the three contractions consume
extern "C" int t4r_head_split_prepare(void* stream, const float* X, long ldx, int N, int D, int V, void* ws) {
    if (N <= 0) return 0;
    T4R_CHECK_ARG(t4r_head_split_supported(D) && X && ws, "head_split_prepare: unsupported width or null pointer");
    T4R_CHECK_ARG(aligned16(X) && ldx % 4 == 0 && aligned16(ws), "head_split_prepare: X must be 16-byte aligned with a pitch multiple of 4");
    const HeadWs w = head_ws(N, V, D);
    hipStream_t st = (hipStream_t)stream;
    u32x4* xa = reinterpret_cast<u32x4*>((char*)ws + w.xa);
    u32x4* xt = reinterpret_cast<u32x4*>((char*)ws + w.xt);
    if (head_fwd_fp16x2()) {
        unsigned* amax = reinterpret_cast<unsigned*>((char*)ws + w.scales);
        if (hipMemsetAsync(amax, 0, 8, st) != hipSuccess) { t4r_set_error("head_split_prepare: memset failed"); return -1; }
        const long n4 = (long)N * (D / 4);
        hipLaunchKernelGGL(amax_kernel, dim3((unsigned)min(256L, (n4 + 255) / 256)), dim3(256), 0, st, X, ldx, (long)N, D, amax);
        if (head_dw_fp16x2()) {     // d W may run in either form (it needs the forward's column maxima): all three images, one launch
            u32x4* xth = reinterpret_cast<u32x4*>((char*)ws + w.xth);
            T4R_NB_SWITCH(D, hipLaunchKernelGGL(split_x_images_kernel<NB>, dim3(w.nblk, 3), dim3(256), 0, st, X, ldx, N, xa, xt, xth, amax));
            T4R_LAUNCH_CHECK();
            return 0;
        }
        T4R_NB_SWITCH(D, hipLaunchKernelGGL((split_mk_kernel<NB, true>), dim3(w.nblk), dim3(256), 0, st, X, ldx, N, xa, amax));
    } else {
        T4R_NB_SWITCH(D, hipLaunchKernelGGL(split_mk_kernel<NB>, dim3(w.nblk), dim3(256), 0, st, X, ldx, N, xa));
    }
    T4R_NB_SWITCH(D, hipLaunchKernelGGL(split_km_kernel<NB>, dim3(w.nblk), dim3(256), 0, st, X, ldx, N, xt));
    T4R_LAUNCH_CHECK();
    return 0;
}

// the one more image of X the RECOMPUTING head's d W reads (accumulator-ordered k-major blocks, split_kmp_kernel): after
// t4r_head_split_prepare on the same X and workspace, before t4r_head_split_ce.  Its own entry so that the materialised head
// (the default where the scores fit) does not pay a launch for it on the critical stream.
extern "C" int t4r_head_split_recompute_supported(int D);
extern "C" int t4r_head_split_prepare_rc(void* stream, const float* X, long ldx, int N, int D, int V, void* ws) {
    if (N <= 0) return 0;
    T4R_CHECK_ARG(t4r_head_split_recompute_supported(D) && X && ws, "head_split_prepare_rc: unsupported (the two-way fp16 forms must be on) or null pointer");
    T4R_CHECK_ARG(aligned16(X) && ldx % 4 == 0 && aligned16(ws), "head_split_prepare_rc: X must be 16-byte aligned with a pitch multiple of 4");
    const HeadWs w = head_ws(N, V, D);
    unsigned* amax = reinterpret_cast<unsigned*>((char*)ws + w.scales);
    u32x4* xtp = reinterpret_cast<u32x4*>((char*)ws + w.xtp);
    T4R_NB_SWITCH(D, hipLaunchKernelGGL((split_kmp_kernel<NB, true>), dim3(w.nblk), dim3(256), 0, (hipStream_t)stream, X, ldx, N, xtp, amax));
    T4R_LAUNCH_CHECK();
    return 0;
}

// C[N, V] = alpha * X @ W^T from the prepared workspace
extern "C" int t4r_head_split_logits(void* stream, void* ws, const float* W, long ldw, float* C, long ldc, int N,
                                     int V, int D, float alpha, void* note) {
    if (N <= 0 || V <= 0) return 0;
    T4R_CHECK_ARG(t4r_head_split_supported(D) && W && C && ws, "head_split_logits: unsupported width or null pointer");
    T4R_CHECK_ARG(aligned16(W) && ldw % 4 == 0, "head_split_logits: W must be 16-byte aligned with a pitch multiple of 4");
    const HeadWs w = head_ws(N, V, D);
    static int per_env = -1;
    if (per_env < 0) { const char* e = t4r_exp_getenv("T4R_HEAD_ROWS_PER_WG"); per_env = e ? atoi(e) : 12; }
    const int blk_per = max(1, min(w.nblk, per_env));
    const int rs = (w.nblk + blk_per - 1) / blk_per;
    const u32x4* xa = reinterpret_cast<const u32x4*>((const char*)ws + w.xa);
    dim3 grid((V + 127) / 128, rs);
    if (head_fwd_fp16x2()) {
        unsigned* amax = reinterpret_cast<unsigned*>((char*)ws + w.scales);
        if (head_w_amax((hipStream_t)stream, W, ldw, V, D, amax + 1)) return -1;
        if (note) *note_of(note) = FwdNote{W, C, V, V, N, 0, 0, 0};
        T4R_NB_SWITCH(D, hipLaunchKernelGGL((head_logits_split_kernel<NB, true>), grid, dim3(256), 0, (hipStream_t)stream, xa, W,
                                            ldw, C, ldc, N, V, alpha, w.nblk, blk_per, amax));
    } else {
        T4R_NB_SWITCH(D, hipLaunchKernelGGL(head_logits_split_kernel<NB>, grid, dim3(256), 0, (hipStream_t)stream, xa, W, ldw, C,
                                            ldc, N, V, alpha, w.nblk, blk_per));
    }
    T4R_LAUNCH_CHECK();
    return 0;
}

// logits + mean cross-entropy in one pass over the vocabulary: C as t4r_head_split_logits, plus loss_rows [N], lse [N]
// and (optional) the mean loss -- replaces t4r_softmax_ce_fwd on the materialised logits
int t4r_mean_launch(hipStream_t stream, const float* x, int n, float* out);      // head.hip
extern "C" int t4r_head_split_logits_ce(void* stream, void* ws, const float* W, long ldw, float* C, long ldc,
                                        const long* labels, float* loss_rows, float* lse, float* loss_mean, int N, int V,
                                        int D, float alpha, float label_smoothing, void* note) {
    hipStream_t st = (hipStream_t)stream;
    if (N <= 0 || V <= 0) return loss_mean ? t4r_mean_launch(st, loss_rows, 0, loss_mean) : 0;
    T4R_CHECK_ARG(t4r_head_split_supported(D) && W && C && ws && (!labels || (loss_rows && lse)), "head_split_logits_ce: unsupported width or null pointer");
    T4R_CHECK_ARG(aligned16(W) && ldw % 4 == 0, "head_split_logits_ce: W must be 16-byte aligned with a pitch multiple of 4");
    const HeadWs w = head_ws(N, V, D);
    static int per_env = -1;
    if (per_env < 0) { const char* e = t4r_exp_getenv("T4R_HEAD_ROWS_PER_WG"); per_env = e ? atoi(e) : 12; }
    const int blk_per = max(1, min(w.nblk, per_env));
    const int rs = (w.nblk + blk_per - 1) / blk_per;
    const u32x4* xa = reinterpret_cast<const u32x4*>((const char*)ws + w.xa);
    float* sm = reinterpret_cast<float*>((char*)ws + w.stats);
    float* ss = sm + (long)w.ntile * N;
    float* stt = label_smoothing > 0.f ? ss + (long)w.ntile * N : nullptr;
    const int vec_ok = aligned16(C) && ldc % 4 == 0;
    dim3 grid(8 * ((w.ntile + 7) / 8) * rs);
    if (head_fwd_fp16x2()) {
        unsigned* amax = reinterpret_cast<unsigned*>((char*)ws + w.scales);
        if (head_w_amax(st, W, ldw, V, D, amax + 1)) return -1;
        float* colmax = nullptr;
        if (head_dw_fp16x2() && rs == w.rsplit && vec_ok) colmax = reinterpret_cast<float*>((char*)ws + w.colmax);
        if (note) *note_of(note) = FwdNote{W, C, V, V, N, colmax != nullptr, 0, 0};
        T4R_NB_SWITCH(D, hipLaunchKernelGGL(head_logits_ce_h_kernel<NB>, grid, dim3(256), 0, st, xa, W, ldw, C, ldc, N, V,
                                            alpha, w.nblk, blk_per, vec_ok, sm, ss, stt, w.ntile, rs, amax, colmax, w.vpad));
    } else {
        T4R_NB_SWITCH(D, hipLaunchKernelGGL(head_logits_ce_kernel<NB>, grid, dim3(256), 0, st, xa, W, ldw, C, ldc, N, V, alpha,
                                            w.nblk, blk_per, vec_ok, sm, ss, stt, w.ntile, rs));
    }
    if (labels)       // labels == NULL: the product and its per-tile statistics only (timing the dominant kernel alone)
        hipLaunchKernelGGL(head_ce_finalize_kernel, dim3((N + 31) / 32), dim3(1024), 0, st, sm, ss, stt, w.ntile, N, V, C, ldc,
                           labels, label_smoothing, loss_rows, lse);
    T4R_LAUNCH_CHECK();
    return labels && loss_mean ? t4r_mean_launch(st, loss_rows, N, loss_mean) : 0;
}

// d W[Vc, D] (+)= alpha * dlogits^T @ X;  logits holds the columns [yoff, yoff + Vc) of the [N, V] problem
extern "C" int t4r_head_split_dw(void* stream, void* ws, const float* logits, long ld, const float* lse,
                                 const long* labels, const float* grad_out, float label_smoothing, float* dW, long lddw,
                                 int N, int Vc, int V, int yoff, int D, float alpha, int accumulate, void* note_p) {
    if (N <= 0 || Vc <= 0) return 0;
    T4R_CHECK_ARG(t4r_head_split_supported(D) && logits && lse && labels && dW && ws, "head_split_dw: unsupported width or null pointer");
    const HeadWs w = head_ws(N, V, D);
    const u32x4* xt = reinterpret_cast<const u32x4*>((const char*)ws + w.xt);
    FwdNote* note = note_of(note_p);
    const bool have_cm = note && note->colmax && note->logits == logits && note->V == V && note->N == N && Vc == V && yoff == 0;
    if (note) note->dw_form = (head_fwd_fp16x2() && head_dw_fp16x2() && have_cm) ? 2 : 1;
    if (head_fwd_fp16x2() && head_dw_fp16x2() && have_cm) {
        hipStream_t st = (hipStream_t)stream;
        const unsigned* amax = reinterpret_cast<const unsigned*>((const char*)ws + w.scales);
        float* lse_min = reinterpret_cast<float*>(const_cast<char*>((const char*)ws) + w.scales) + 2;
        unsigned char* islab = reinterpret_cast<unsigned char*>(const_cast<char*>((const char*)ws) + w.islab);
        hipLaunchKernelGGL(head_dw_aux_kernel, dim3(1), dim3(1024), 0, st, lse, labels, N, yoff, Vc, lse_min, islab, w.vpad);
        DwAux aux{reinterpret_cast<const float*>((const char*)ws + w.colmax), lse_min, islab, w.vpad, note->cm_rows > 0 ? note->cm_rows : w.rsplit};
        const u32x4* xth = reinterpret_cast<const u32x4*>((const char*)ws + w.xth);
        // PD = 1, non-temporal logits (round 6 A/B, profiles/r06_b_head_dw_variants.txt: two / three / four blocks in flight are
        // 2-7 % SLOWER alone -- the launch is not waiting for its loads --, the non-temporal form is the only one ahead: -0.2 %)
        T4R_NB_SWITCH(D, hipLaunchKernelGGL((head_dw_split_kernel<NB, true, 1, true>), dim3((Vc + 127) / 128), dim3(256), 0, st, logits,
                                            ld, lse, labels, grad_out, xth, dW, lddw, N, Vc, V, yoff, label_smoothing, alpha,
                                            accumulate, w.nblk, amax, aux));
    } else {
        T4R_NB_SWITCH(D, hipLaunchKernelGGL(head_dw_split_kernel<NB>, dim3((Vc + 127) / 128), dim3(256), 0, (hipStream_t)stream,
                                            logits, ld, lse, labels, grad_out, xt, dW, lddw, N, Vc, V, yoff, label_smoothing,
                                            alpha, accumulate, w.nblk));
    }
    T4R_LAUNCH_CHECK();
    return 0;
}

// d X[N, D] (+)= alpha * dlogits @ W[yoff : yoff + Vc];  W points at row yoff
extern "C" int t4r_head_split_dx(void* stream, void* ws, const float* logits, long ld, const float* lse,
                                 const long* labels, const float* grad_out, float label_smoothing, const float* W, long ldw,
                                 float* dX, long lddx, int N, int Vc, int V, int yoff, int D, float alpha, int accumulate,
                                 void* note_p) {
    if (N <= 0 || Vc <= 0) return 0;
    T4R_CHECK_ARG(t4r_head_split_supported(D) && logits && lse && labels && dX && W && ws, "head_split_dx: unsupported width or null pointer");
    T4R_CHECK_ARG(aligned16(logits) && ld % 4 == 0 && ld >= 8 && aligned16(dX) && lddx % 4 == 0,
                  "head_split_dx: logits / dX must be 16-byte aligned with pitches multiple of 4");
    const HeadWs w = head_ws(N, V, D);
    hipStream_t st = (hipStream_t)stream;
    u32x4* wt = reinterpret_cast<u32x4*>((char*)ws + w.wt);
    float* part = reinterpret_cast<float*>((char*)ws + w.part);
    const int nkt = (Vc + 31) / 32, row_tiles = (N + 127) / 128;
    const bool hs = head_fwd_fp16x2();
    unsigned* amax = reinterpret_cast<unsigned*>((char*)ws + w.scales) + 1;     // max |W| of THIS call's rows (a chunk of the table)
    if (hs) {
        FwdNote* note = note_of(note_p);
        const bool same = note && note->W == W && note->Vw == Vc;
        if (!same) {
            if (head_w_amax(st, W, ldw, Vc, D, amax)) return -1;
            if (note) { note->W = W; note->Vw = Vc; }     // the max |W| word now describes THIS table slice
        }
        T4R_NB_SWITCH(D, hipLaunchKernelGGL((split_km_kernel<NB, true>), dim3(nkt), dim3(256), 0, st, W, ldw, Vc, wt, amax));
    } else {
        T4R_NB_SWITCH(D, hipLaunchKernelGGL(split_km_kernel<NB>, dim3(nkt), dim3(256), 0, st, W, ldw, Vc, wt));
    }
    const int target = head_dx_target();
    int splits = max(1, min(min(w.max_split, nkt / 8), target / row_tiles));
    const int kt_per = (nkt + splits - 1) / splits;
    splits = (nkt + kt_per - 1) / kt_per;          // every split owns at least one k-tile
    if (hs) {
        T4R_NB_SWITCH(D, hipLaunchKernelGGL((head_dx_split_kernel<NB, true>), dim3(row_tiles * 8 * ((splits + 7) / 8)), dim3(256),
                                            0, st, logits, ld, lse, labels, grad_out, wt, part, N, Vc, V, yoff, label_smoothing,
                                            alpha, nkt, kt_per, row_tiles, amax));
    } else {
        T4R_NB_SWITCH(D, hipLaunchKernelGGL(head_dx_split_kernel<NB>, dim3(row_tiles * 8 * ((splits + 7) / 8)), dim3(256), 0, st,
                                            logits, ld, lse, labels, grad_out, wt, part, N, Vc, V, yoff, label_smoothing, alpha,
                                            nkt, kt_per, row_tiles));
    }
    const long nd4 = (long)N * D / 4;
    hipLaunchKernelGGL(head_dx_reduce_kernel, dim3((unsigned)((nd4 + 255) / 256)), dim3(256), 0, st, part, splits, nd4, D / 4,
                       dX, lddx, accumulate);
    T4R_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// The recomputing head (see head_dw_rc_kernel): cross-entropy WITHOUT a logits tensor, and its two backward products.
extern "C" int t4r_head_split_recompute_supported(int D) { return t4r_head_split_supported(D) && head_recompute_on(); }

// loss_rows [N], lse [N] (+ the mean loss) of softmax(alpha X W^T) against labels, from the prepared workspace: statistics
// per 128-item tile merged by a small kernel, each row's label logit captured by the lane that holds it.  Nothing of size
// [N, V] is written; predictions, when somebody wants them, are t4r_head_split_logits on the same workspace.
extern "C" int t4r_head_split_ce(void* stream, void* ws, const float* W, long ldw, const long* labels, float* loss_rows,
                                 float* lse, float* loss_mean, int N, int V, int D, float alpha, float label_smoothing, void* note) {
    hipStream_t st = (hipStream_t)stream;
    if (N <= 0 || V <= 0) return loss_mean ? t4r_mean_launch(st, loss_rows, 0, loss_mean) : 0;
    T4R_CHECK_ARG(t4r_head_split_recompute_supported(D) && W && ws && labels && loss_rows && lse, "head_split_ce: unsupported (the two-way fp16 form must be on) or null pointer");
    T4R_CHECK_ARG(aligned16(W) && ldw % 4 == 0, "head_split_ce: W must be 16-byte aligned with a pitch multiple of 4");
    const HeadWs w = head_ws(N, V, D);
    const int blk_per = max(1, min(w.nblk, head_rows_per_wg()));
    const int rs = (w.nblk + blk_per - 1) / blk_per;
    T4R_CHECK_ARG(rs == w.rsplit, "head_split_ce: row split mismatch");
    const u32x4* xa = reinterpret_cast<const u32x4*>((const char*)ws + w.xa);
    float* sm = reinterpret_cast<float*>((char*)ws + w.stats);
    float* ss = sm + (long)w.ntile * N;
    float* stt = label_smoothing > 0.f ? ss + (long)w.ntile * N : nullptr;
    unsigned* amax = reinterpret_cast<unsigned*>((char*)ws + w.scales);
    float* colmax = reinterpret_cast<float*>((char*)ws + w.colmax);
    float* zlab = reinterpret_cast<float*>((char*)ws + w.zlab);
    if (head_w_amax(st, W, ldw, V, D, amax + 1)) return -1;
    if (note) *note_of(note) = FwdNote{W, nullptr, V, V, N, 1, 0, 0};
    dim3 grid(8 * ((w.ntile + 7) / 8) * rs);
    T4R_NB_SWITCH(D, hipLaunchKernelGGL(head_ce_stats_h_kernel<NB>, grid, dim3(256), 0, st, xa, W, ldw, N, V, alpha, w.nblk, blk_per, sm, ss,
                                        stt, w.ntile, rs, amax, colmax, w.vpad, labels, zlab));
    hipLaunchKernelGGL(head_ce_finalize_kernel, dim3((N + 31) / 32), dim3(1024), 0, st, sm, ss, stt, w.ntile, N, V, (const float*)nullptr,
                       0L, labels, label_smoothing, loss_rows, lse, (const float*)zlab);
    T4R_LAUNCH_CHECK();
    return loss_mean ? t4r_mean_launch(st, loss_rows, N, loss_mean) : 0;
}

// d W[V, D] (+)= alpha * dlogits^T @ X with the score tiles recomputed (same workspace and note as t4r_head_split_ce)
extern "C" int t4r_head_split_dw_rc(void* stream, void* ws, const float* W, long ldw, const float* lse, const long* labels,
                                    const float* grad_out, float label_smoothing, float* dW, long lddw, int N, int V, int D,
                                    float alpha, int accumulate, void* note_p) {
    if (N <= 0 || V <= 0) return 0;
    T4R_CHECK_ARG(t4r_head_split_recompute_supported(D) && W && lse && labels && dW && ws, "head_split_dw_rc: unsupported or null pointer");
    FwdNote* note = note_of(note_p);
    T4R_CHECK_ARG(note && note->colmax && note->W == W && note->V == V && note->N == N, "head_split_dw_rc: the note of this workspace's t4r_head_split_ce is required");
    const HeadWs w = head_ws(N, V, D);
    hipStream_t st = (hipStream_t)stream;
    const unsigned* amax = reinterpret_cast<const unsigned*>((const char*)ws + w.scales);
    float* lse_min = reinterpret_cast<float*>((char*)ws + w.scales) + 2;
    unsigned char* islab = reinterpret_cast<unsigned char*>((char*)ws + w.islab);
    hipLaunchKernelGGL(head_dw_aux_kernel, dim3(1), dim3(1024), 0, st, lse, labels, N, 0, V, lse_min, islab, w.vpad);
    DwAux aux{reinterpret_cast<const float*>((const char*)ws + w.colmax), lse_min, islab, w.vpad, w.rsplit};
    const u32x4* xa = reinterpret_cast<const u32x4*>((const char*)ws + w.xa);
    const u32x4* xtp = reinterpret_cast<const u32x4*>((const char*)ws + w.xtp);
    note->dw_form = 3;
    T4R_NB_SWITCH(D, hipLaunchKernelGGL(head_dw_rc_kernel<NB>, dim3((V + 127) / 128), dim3(256), 0, st, xa, xtp, W, ldw, lse, labels,
                                        grad_out, dW, lddw, N, V, label_smoothing, alpha, accumulate, w.nblk, amax, aux));
    T4R_LAUNCH_CHECK();
    return 0;
}

// d X[N, D] (+)= alpha * dlogits @ W with the score tiles recomputed.  X: the rows t4r_head_split_prepare was given.
extern "C" int t4r_head_split_dx_rc(void* stream, void* ws, const float* X, long ldx, const float* W, long ldw, const float* lse,
                                    const long* labels, const float* grad_out, float label_smoothing, float* dX, long lddx,
                                    int N, int V, int D, float alpha, int accumulate, void* note_p) {
    if (N <= 0 || V <= 0) return 0;
    T4R_CHECK_ARG(t4r_head_split_recompute_supported(D) && X && W && lse && labels && dX && ws, "head_split_dx_rc: unsupported or null pointer");
    T4R_CHECK_ARG(aligned16(X) && ldx % 4 == 0 && aligned16(W) && ldw % 4 == 0 && aligned16(dX) && lddx % 4 == 0,
                  "head_split_dx_rc: X / W / dX must be 16-byte aligned with pitches multiple of 4");
    FwdNote* note = note_of(note_p);
    T4R_CHECK_ARG(note && note->W == W && note->Vw == V && note->N == N, "head_split_dx_rc: the note of this workspace's t4r_head_split_ce is required");
    const HeadWs w = head_ws(N, V, D);
    hipStream_t st = (hipStream_t)stream;
    const unsigned* amax = reinterpret_cast<const unsigned*>((const char*)ws + w.scales);
    u32x4* wtp = reinterpret_cast<u32x4*>((char*)ws + w.wt);
    u32x4* wa = reinterpret_cast<u32x4*>((char*)ws + w.wa);
    float* part = reinterpret_cast<float*>((char*)ws + w.part);
    const int nkt = w.nkt, row_tiles = (N + 127) / 128;
    // the table in both images, positioned by the max |W| the forward left (same table, same rows)
    T4R_NB_SWITCH(D, hipLaunchKernelGGL((split_mk_kernel<NB, true>), dim3(nkt), dim3(256), 0, st, W, ldw, V, wa, amax + 1));
    T4R_NB_SWITCH(D, hipLaunchKernelGGL((split_kmp_kernel<NB, true>), dim3(nkt), dim3(256), 0, st, W, ldw, V, wtp, amax + 1));
    const int target = head_dx_target();
    int splits = max(1, min(min(w.max_split, nkt / 8), target / row_tiles));
    const int kt_per = (nkt + splits - 1) / splits;
    splits = (nkt + kt_per - 1) / kt_per;
    T4R_NB_SWITCH(D, hipLaunchKernelGGL(head_dx_rc_kernel<NB>, dim3(row_tiles * 8 * ((splits + 7) / 8)), dim3(256), 0, st, X, ldx, wa, wtp,
                                        lse, labels, grad_out, part, N, V, label_smoothing, alpha, nkt, kt_per, row_tiles, amax));
    const long nd4 = (long)N * D / 4;
    hipLaunchKernelGGL(head_dx_reduce_kernel, dim3((unsigned)((nd4 + 255) / 256)), dim3(256), 0, st, part, splits, nd4, D / 4, dX, lddx,
                       accumulate);
    T4R_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// The one-pass forward (round 5, see head_fwd_dx_kernel): logits C [N, V], loss rows, lse, the mean loss AND
// dX [N, D] = d (mean loss) / d X for grad_out = 1 -- the caller's backward multiplies by its upstream gradient and runs only
// t4r_head_split_dw (same workspace and note).  X: the rows t4r_head_split_prepare was given.  wsum: column sums of W [D]
// (needed when label_smoothing > 0, else NULL).
// The next t4r_head_split_logits_ce_dx of this thread on table W takes max |W| from `part[0 .. n)` (per-workgroup maxima written by
// t4r_adam_step_amax over exactly W's elements, AFTER which nothing else has written W -- the caller's promise) instead of
// computing it.  Consumed (cleared) by that call, whatever table it is given.
static thread_local const float* g_w_amax_part = nullptr;
static thread_local const float* g_w_amax_W = nullptr;
static thread_local int g_w_amax_n = 0;
extern "C" void t4r_head_split_w_amax_hint(const float* W, const float* part, int n) {
    g_w_amax_W = (part && n > 0) ? W : nullptr; g_w_amax_part = g_w_amax_W ? part : nullptr; g_w_amax_n = g_w_amax_W ? n : 0;
}
extern "C" int t4r_head_split_fdx_supported(int D) { return t4r_head_split_supported(D) && head_fdx_on(); }
extern "C" int t4r_head_split_logits_ce_dx(void* stream, void* ws, const float* X, long ldx, const float* W, long ldw, float* C,
                                           long ldc, const long* labels, float* loss_rows, float* lse, float* loss_mean,
                                           float* dX, long lddx, const float* wsum, int N, int V, int D, float alpha,
                                           float label_smoothing, void* note) {
    hipStream_t st = (hipStream_t)stream;
    if (N <= 0 || V <= 0) return loss_mean ? t4r_mean_launch(st, loss_rows, 0, loss_mean) : 0;
    // labels == NULL: the dominant kernel alone (no table maximum, no image cut, no finalize: the workspace holds them from
    // a previous full call on the same X and W) -- what bench.py times for its roofline, as t4r_head_split_logits_ce does
    const bool kernel_only = labels == nullptr;
    T4R_CHECK_ARG(t4r_head_split_fdx_supported(D) && X && W && C && ws && (kernel_only || (loss_rows && lse && dX)),
                  "head_split_logits_ce_dx: unsupported (the two-way fp16 form must be on) or null pointer");
    T4R_CHECK_ARG(aligned16(X) && ldx % 4 == 0 && aligned16(W) && ldw % 4 == 0 && aligned16(dX) && lddx % 4 == 0,
                  "head_split_logits_ce_dx: X / W / dX must be 16-byte aligned with pitches multiple of 4");
    T4R_CHECK_ARG(label_smoothing <= 0.f || wsum, "head_split_logits_ce_dx: label smoothing needs the column sums of W");
    const HeadWs w = head_ws(N, V, D);
    unsigned* amax = reinterpret_cast<unsigned*>((char*)ws + w.scales);
    // the table's maximum: from the optimizer's partials when the caller announced them for THIS table (t4r_head_split_w_amax_hint,
    // consumed here), else a memset + a pass over the table
    const float* w_part = g_w_amax_W == W ? g_w_amax_part : nullptr;
    const int w_npart = g_w_amax_n;
    g_w_amax_part = nullptr; g_w_amax_W = nullptr; g_w_amax_n = 0;
    if (!kernel_only && !w_part && head_w_amax(st, W, ldw, V, D, amax + 1)) return -1;
    u32x4* wa = reinterpret_cast<u32x4*>((char*)ws + w.wa);
    u32x4* wtp = reinterpret_cast<u32x4*>((char*)ws + w.wt);
    if (!kernel_only)
        T4R_NB_SWITCH(D, hipLaunchKernelGGL(split_w_images_kernel<NB>, dim3(w.nkt, 2), dim3(256), 0, st, W, ldw, V, wa, wtp, amax + 1,
                                            w_part, w_npart));
    const int row_tiles = (N + 127) / 128;
    // one residency of the chip: two 256-thread workgroups per CU (66 KB of LDS each), every workgroup the same number of tiles
    static int target = -1;
    if (target < 0) { const char* e = t4r_exp_getenv("T4R_HEAD_FDX_WGS"); target = e ? max(1, atoi(e)) : 512; }
    int splits = max(1, min(min(64, w.max_split), min(max(1, w.nkt / 8), target / row_tiles)));
    const int kt_per = (w.nkt + splits - 1) / splits;
    splits = (w.nkt + kt_per - 1) / kt_per;
    float* sm = reinterpret_cast<float*>((char*)ws + w.stats);
    float* ss = sm + (long)w.ntile * N;
    float* stt = label_smoothing > 0.f ? ss + (long)w.ntile * N : nullptr;
    float* part = reinterpret_cast<float*>((char*)ws + w.part);
    const int vec_ok = aligned16(C) && ldc % 4 == 0;
    float* colmax = (head_dw_fp16x2() && vec_ok) ? reinterpret_cast<float*>((char*)ws + w.colmax) : nullptr;
    if (note) *note_of(note) = FwdNote{W, C, V, V, N, colmax != nullptr, 0, row_tiles};
    const int n_wg = splits * row_tiles, per_xcd = (n_wg + 7) / 8;
    if (stt) {
        T4R_NB_SWITCH(D, hipLaunchKernelGGL((head_fwd_dx_kernel<NB, true>), dim3(8 * per_xcd), dim3(256), 0, st, X, ldx, wa, wtp, C, ldc, vec_ok,
                                            part, sm, ss, stt, colmax, w.vpad, N, V, alpha, w.nkt, kt_per, row_tiles, n_wg, per_xcd, amax));
    } else {
        T4R_NB_SWITCH(D, hipLaunchKernelGGL((head_fwd_dx_kernel<NB, false>), dim3(8 * per_xcd), dim3(256), 0, st, X, ldx, wa, wtp, C, ldc, vec_ok,
                                            part, sm, ss, stt, colmax, w.vpad, N, V, alpha, w.nkt, kt_per, row_tiles, n_wg, per_xcd, amax));
    }
    if (!kernel_only)
        hipLaunchKernelGGL(head_fdx_finalize_kernel, dim3((N + 7) / 8), dim3(256), 0, st, sm, ss, stt, part, splits, N, V, D, C, ldc, W, ldw,
                           labels, wsum, label_smoothing, alpha, amax, loss_rows, lse, dX, lddx);
    T4R_LAUNCH_CHECK();
    return (loss_mean && !kernel_only) ? t4r_mean_launch(st, loss_rows, N, loss_mean) : 0;
}
