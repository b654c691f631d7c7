// Token-tile-stationary fused kernels of the XLNet layer (round 3): one workgroup owns a tile of 16*R token rows
// and runs a whole sub-block of the layer on it, activations never leaving the CU between the GEMMs.
//
//   xlnet_ff_fwd_kernel   h1 -> FF1 + bias + GELU(erf) + dropout -> FF2 + bias + dropout + residual -> LayerNorm
//                         (HF modeling_xlnet.py XLNetFeedForward.forward :297-305; five launches of the chain in
//                          xlnet_layer.hip: two GEMMs with epilogues, LayerNorm -- and no second trip of the
//                          [T, 4D] activation through HBM)
//   xlnet_ff_bwd_kernel   d h2 -> LayerNorm backward -> d ffout -> FF2 dX -> GELU' / dropout -> FF1 dX + residual
//                         (+ per-workgroup partial sums of d gamma, d beta, d b2, d b1; the weight gradients stay
//                          separate token-reducing GEMMs fed by the d ffout / d pre rows this kernel writes)
//
// Arithmetic: fp32-accurate products on the BF16 matrix cores -- every operand is cut into three bf16 planes
// (x = hi + mid + lo exactly, round-to-nearest cuts) and the six largest partial products are accumulated in fp32 by
// v_mfma_f32_16x16x32_bf16 (the scheme of gemm_kernel.h PREC 1 / head_split.hip: error at the level of an fp32 FMA chain).
// The WEIGHTS are cut once per layer call by split_planes_kernel (also in transposed order for the backward), the token
// tile once when it enters LDS, the activation chunk once in the producing epilogue.
// Why not the fp32 matrix instruction (the first version of this file used v_mfma_f32_16x16x4_f32): measured on this chip
// (tools/mfma_valu_overlap.hip) VALU work does NOT issue in the shadow of matrix instructions of the same SIMD -- fp32 MFMA
// + VALU = the sum of both, bf16 MFMA + VALU nearly so -- and the GELU / Philox epilogue here is ~40 % of the fp32 MFMA time:
// 77 us per forward launch however finely the epilogue was interleaved.  So the total of (matrix cycles + VALU cycles) is
// what counts, and six bf16 MFMAs (6 x 16 cycles per 16x16x32) cost 2.7x less than the eight fp32 ones (8 x 32).
//
// The products are formed TRANSPOSED -- the weight is the MFMA "A" operand (m = output feature), the token tile is "B"
// (n = token) -- so that
//   * an accumulator lane holds FOUR CONSECUTIVE FEATURES of ONE token: bias / GELU / Philox mask (one block = four
//     consecutive columns of a row, the indexing of every other kernel here) / residual / stores are all 16-byte;
//   * NW = D/16 waves split the FEATURE dimension: every wave works on all 16*R tokens, so 20 480 tokens on 256 CUs
//     (80 per CU, R = 5) load every SIMD equally -- splitting tokens over waves cannot (20 tokens per SIMD);
//   * each weight tile is fetched once per workgroup (L2-resident: 1.5 MB of planes per layer for 256 workgroups).
// Operand fragments (16x16x32: lane (i = lane & 15, kg = lane >> 4) holds k = 32 s + 8 kg .. + 7) are one 16-byte load
// per plane from a k-contiguous row: global for the weight planes, LDS for the token planes.
// LDS: token planes [3][16R][D + 16] bf16 (B operand of FF1; the 32-byte row pad makes the 16-byte fragment reads of a
// ds_read_b128 lane group hit 16 distinct bank quads: SQ_LDS_BANK_CONFLICT was 46 % of the LDS cycles with D + 8),
// activation-chunk planes [3][16R][D + 16] (written by the
// FF1 epilogue, B operand of FF2; d_inner = 4 D is walked in four chunks of D, two barriers per chunk).
#include "xlnet_fused.h"

struct FFFwdParams {
    const float *h1, *b1, *b2, *gamma, *beta;
    LayerPlanes w;
    LayerPlanesH wh;       // HS: the two-way fp16 planes and their scales
    float *amax_h1, *amax_act;      // HS, optional: per-workgroup max |h1|, max |act| (operand maxima of the weight gradients)
    float *ffpre, *ffact, *ffout, *mean, *rstd, *hout;     // ffpre / ffact / ffout / mean / rstd: all given (training) or all NULL
    int T;
    float eps;
    DropCfg drop_act, drop_out;
    DropCfg drop_final;    // p > 0: the MODEL's output dropout (HF modeling_xlnet.py:1177) applied to hout in this launch (last layer)
};

// FULL: every row of the tile is a token; TRAIN: the backward's activations are saved and the Philox masks evaluated
// (p = 0 keeps everything); !TRAIN (inference: nothing saved, p = 0): neither.
// HS: both products on the two-way fp16 split (three matrix instructions per k-step instead of six).  Token rows carry their
// own power-of-two scale; the activation rows are positioned by a per-token BOUND known before the first chunk
// (|gelu(x)| <= |x|, |pre[t][f]| <= D max|W1| max|h1[t]| + max|b1|, the dropout factor on top): one scale for all four
// chunks of a token, as their products share an accumulator.  The bound is loose by one to two orders of magnitude, which
// costs nothing: fp16 keeps its 11 + 11 bits over 2^17 of range below the bound.
template <int D, int R, bool FULL, bool TRAIN, bool HS>
__device__ __forceinline__ void ff_fwd_body(const FFFwdParams& p, uint16_t* smem) {
    constexpr int NW = D / 16, NT = NW * 64, RT = 16 * R, PH = D + 16, PLN = RT * PH, DI = 4 * D;
    uint16_t* sh_h = smem;                                   // [3][RT][PH] token planes
    uint16_t* sh_a = sh_h + 3 * PLN;                         // [3][RT][PH] activation-chunk planes
    float* sh_red = reinterpret_cast<float*>(sh_a + 3 * PLN);   // [2][NW][RT]
    float* sh_inv_h = reinterpret_cast<float*>(sh_h + 2 * PLN);    // HS: [RT] inverse token scales (in the unused third plane)
    float* sh_sa = reinterpret_cast<float*>(sh_a + 2 * PLN);       // HS: [RT] activation scales | [RT] their inverses
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, g = lane >> 4;
    const long t0 = (long)blockIdx.x * RT;
    float h1max_keep = 0.f;
    if constexpr (HS) {
        const float w1max = 16384.f / p.wh.scale[HS_W1];    // >= max |W1| (the scale puts it into [2^13, 2^14))
        const float b1max = p.wh.scale[HS_B1];
        const float keep = TRAIN ? p.drop_act.inv_keep : 1.f;
        constexpr int G = D / 4;
        float h1max = 0.f;
        for (int i = tid; i < ((RT * G + NT - 1) / NT) * NT; i += NT) {
            const bool live = i < RT * G;
            const int row = live ? i / G : RT - 1, c4 = live ? (i % G) * 4 : 0;
            const long t = FULL ? t0 + row : min(t0 + row, (long)p.T - 1);
            const float4 v = ld4(p.h1 + t * D + c4);
            float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            const float sc = pow2_scale(m);
            if (live) {
                h1max = fmaxf(h1max, m);
                uint32_t w0[2], w1[2];
                cut2h(v.x * sc, v.y * sc, w0);
                cut2h(v.z * sc, v.w * sc, w1);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    *reinterpret_cast<uint2*>(sh_h + pl * PLN + row * PH + c4) = make_uint2(w0[pl], w1[pl]);
                if (c4 == 0) {
                    sh_inv_h[row] = 1.f / sc;
                    const float sa = pow2_scale(((float)D * w1max * m + b1max) * keep);
                    sh_sa[row] = sa;
                    sh_sa[RT + row] = 1.f / sa;
                }
            }
        }
        h1max_keep = h1max;
    } else
    for (int i = tid; i < RT * (D / 4); i += NT) {
        const int row = i / (D / 4), c4 = (i % (D / 4)) * 4;
        const long t = FULL ? t0 + row : min(t0 + row, (long)p.T - 1);
        const float4 v = ld4(p.h1 + t * D + c4);
        uint32_t w0[3], w1[3];
        cut3(v.x, v.y, w0);
        cut3(v.z, v.w, w1);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<uint2*>(sh_h + pl * PLN + row * PH + c4) = make_uint2(w0[pl], w1[pl]);
    }
    __syncthreads();
    f32x4 acc2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc2[r] = zero4();
    const int boff = n * PH + 8 * g;                          // this lane's B-fragment offset inside a token block
    const long wpl = (long)DI * D;                            // plane stride of every weight-plane matrix (4 D * D)
    const uint16_t* w1p = (HS ? p.wh.W1p : p.w.W1p) + (long)(16 * w + n) * D + 8 * g;       // + c D D: chunk c
    const uint16_t* w2p = (HS ? p.wh.W2p : p.w.W2p) + (long)(16 * w + n) * DI + 8 * g;      // + c D
    AFrag<D> a1, a2;
    AFragH<D> h1f, h2f;
    float actmax = 0.f;            // HS, training: running max |act| of this lane (operand maximum for the W2 weight gradient)
    float is1[R], is2[R];          // HS: what turns an accumulator of FF1 / FF2 back into values, per token block
    if constexpr (HS) {
        const float iw1 = 1.f / p.wh.scale[HS_W1], iw2 = 1.f / p.wh.scale[HS_W2];
#pragma unroll
        for (int r = 0; r < R; ++r) { is1[r] = sh_inv_h[r * 16 + n] * iw1; is2[r] = sh_sa[RT + r * 16 + n] * iw2; }
        load_a2h<D>(h1f, w1p, wpl);
    } else {
        load_a3<D>(a1, w1p, wpl);
    }
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        f32x4 acc1[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc1[r] = zero4();
        if constexpr (HS) {
            product3h<D, R, PH>(h1f, sh_h + boff, PLN, acc1);
            load_a2h<D>(h2f, w2p + c * D, wpl);
#pragma unroll
            for (int r = 0; r < R; ++r) { acc1[r][0] *= is1[r]; acc1[r][1] *= is1[r]; acc1[r][2] *= is1[r]; acc1[r][3] *= is1[r]; }
        } else {
#if T4R_FF_PREFETCH
        load_a3<D>(a2, w2p + c * D, wpl);                     // FF2's fragment: in flight under FF1 and the epilogue
        product3<D, R, PH>(a1, sh_h + boff, PLN, acc1);
        load_a3<D>(a1, w1p + (long)min(c + 1, 3) * D * D, wpl);   // next chunk's FF1 fragment: in flight under the epilogue and FF2
#else
        product3<D, R, PH>(a1, sh_h + boff, PLN, acc1);
        load_a3<D>(a2, w2p + c * D, wpl);                     // in flight under the epilogue (a1 is dead: no extra registers)
#endif
        }
        // epilogue of chunk c: lane = (token r*16 + n, features d0 .. d0 + 3 of d_inner)
        const int d0 = c * D + 16 * w + 4 * g;
        const float4 bias = ld4(p.b1 + d0);
        if (c > 0) __syncthreads();                           // every wave has finished reading the previous chunk's planes
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long t = t0 + r * 16 + n;
            float4 v = make_float4(acc1[r][0] + bias.x, acc1[r][1] + bias.y, acc1[r][2] + bias.z, acc1[r][3] + bias.w);
            const bool in = FULL || t < p.T;
            if (TRAIN && in) st4(p.ffpre + t * DI + d0, v);
            v = make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
            if (TRAIN) {
                const float4 m = drop_scale4(p.drop_act, (unsigned long long)t * DI + d0);
                v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
            }
            if (TRAIN && in) st4(p.ffact + t * DI + d0, v);
            if constexpr (HS) {
                if (TRAIN && in) actmax = fmaxf(actmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                const float sa = sh_sa[r * 16 + n];
                uint32_t w0[2], w1[2];
                cut2h(v.x * sa, v.y * sa, w0);
                cut2h(v.z * sa, v.w * sa, w1);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    *reinterpret_cast<uint2*>(sh_a + pl * PLN + (r * 16 + n) * PH + 16 * w + 4 * g) = make_uint2(w0[pl], w1[pl]);
            } else {
                uint32_t w0[3], w1[3];
                cut3(v.x, v.y, w0);
                cut3(v.z, v.w, w1);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    *reinterpret_cast<uint2*>(sh_a + pl * PLN + (r * 16 + n) * PH + 16 * w + 4 * g) = make_uint2(w0[pl], w1[pl]);
            }
        }
        __syncthreads();
        if constexpr (HS) {
            if (c < 3) load_a2h<D>(h1f, w1p + (long)(c + 1) * D * D, wpl);
            product3h<D, R, PH>(h2f, sh_a + boff, PLN, acc2);
        } else {
#if !T4R_FF_PREFETCH
        if (c < 3) load_a3<D>(a1, w1p + (long)(c + 1) * D * D, wpl);   // next chunk's FF1 fragment, in flight under FF2
#endif
        product3<D, R, PH>(a2, sh_a + boff, PLN, acc2);
        }
    }
    if constexpr (HS) {
#pragma unroll
        for (int r = 0; r < R; ++r) { acc2[r][0] *= is2[r]; acc2[r][1] *= is2[r]; acc2[r][2] *= is2[r]; acc2[r][3] *= is2[r]; }
    }
    // epilogue 2: + b2, dropout, + residual, LayerNorm over the D features of a token (split over the NW waves)
    const int f0 = 16 * w + 4 * g;
    const float4 bias2 = ld4(p.b2 + f0), gam = ld4(p.gamma + f0), bet = ld4(p.beta + f0);
    float4 x[R];
    float sum[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long t = t0 + r * 16 + n;
        const long tc = FULL ? t : min(t, (long)p.T - 1);
        float4 v = make_float4(acc2[r][0] + bias2.x, acc2[r][1] + bias2.y, acc2[r][2] + bias2.z, acc2[r][3] + bias2.w);
        if (TRAIN && (FULL || t < p.T)) st4(p.ffout + t * D + f0, v);
        if (TRAIN) {
            const float4 m = drop_scale4(p.drop_out, (unsigned long long)t * D + f0);
            v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
        }
        const float4 hres = ld4(p.h1 + tc * D + f0);          // the residual in fp32 (L2 hit: this tile was just read)
        v.x += hres.x; v.y += hres.y; v.z += hres.z; v.w += hres.w;
        x[r] = v;
        float s = (v.x + v.y) + (v.z + v.w);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        sum[r] = s;
    }
    if (g == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) sh_red[w * RT + r * 16 + n] = sum[r];
    }
    __syncthreads();
    float mu[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) s += sh_red[ww * RT + r * 16 + n];
        mu[r] = s * (1.0f / D);
        const float dx = x[r].x - mu[r], dy = x[r].y - mu[r], dz = x[r].z - mu[r], dw = x[r].w - mu[r];
        float q = (dx * dx + dy * dy) + (dz * dz + dw * dw);
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        sum[r] = q;
    }
    float* sh_red2 = sh_red + NW * RT;
    if (g == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) sh_red2[w * RT + r * 16 + n] = sum[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long t = t0 + r * 16 + n;
        float q = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) q += sh_red2[ww * RT + r * 16 + n];
        const float rs = rsqrtf(q * (1.0f / D) + p.eps);
        if (FULL || t < p.T) {
            float4 o = make_float4((x[r].x - mu[r]) * rs * gam.x + bet.x, (x[r].y - mu[r]) * rs * gam.y + bet.y,
                                   (x[r].z - mu[r]) * rs * gam.z + bet.z, (x[r].w - mu[r]) * rs * gam.w + bet.w);
            if (TRAIN && p.drop_final.p > 0.f) {      // workgroup-uniform: the last layer of a stack in training mode
                const float4 mf = drop_scale4(p.drop_final, (unsigned long long)t * D + f0);
                o.x *= mf.x; o.y *= mf.y; o.z *= mf.z; o.w *= mf.w;
            }
            st4(p.hout + t * D + f0, o);
            if (TRAIN && w == 0 && g == 0) { p.mean[t] = mu[r]; p.rstd[t] = rs; }
        }
    }
    if constexpr (HS && TRAIN) {
        if (p.amax_h1) {        // workgroup-uniform
            block_amax_to<NW>(h1max_keep, sh_red, p.amax_h1, tid);
            block_amax_to<NW>(actmax, sh_red, p.amax_act, tid);
        }
    }
}

template <int D, int R, bool HS = false>
__global__ __launch_bounds__(D * 4) void xlnet_ff_fwd_kernel(FFFwdParams p) {
    extern __shared__ uint16_t smem16[];
    // whole tiles run the unpredicated body; only the last workgroup of a ragged launch pays for the row guards
    const bool full = (long)(blockIdx.x + 1) * 16 * R <= p.T;
    if (p.ffpre != nullptr) {
        if (full) ff_fwd_body<D, R, true, true, HS>(p, smem16);
        else ff_fwd_body<D, R, false, true, HS>(p, smem16);
    } else {
        if (full) ff_fwd_body<D, R, true, false, HS>(p, smem16);
        else ff_fwd_body<D, R, false, false, HS>(p, smem16);
    }
}

// -------------------------------------------------------------------------------------------------- backward
struct FFBwdParams {
    const float *dy, *ffout, *h1, *mean, *rstd, *gamma, *ffpre;
    LayerPlanes w;
    LayerPlanesH wh;       // HS: the two-way fp16 planes and their scales
    float *amax_dpre, *amax_dfo;    // HS, optional: per-workgroup max |d pre|, max |d ffout|
    float *dh1, *dffout, *dpre;       // [T, D], [T, D], [T, 4D]
    float *partA, *partB;             // per-workgroup partial sums [nWG][3D] (d gamma | d beta | d b2), [nWG][4D] (d b1)
    int T;
    DropCfg drop_act, drop_out;
    DropCfg drop_final;    // p > 0: dy is the gradient of the DROPPED output (model-level site): masked on load
};

// HS: both products on the two-way fp16 split.  The d ffout rows carry their own power-of-two scale (gradient rows of one
// tile differ by orders of magnitude); the d pre rows are positioned by a per-token bound known before the first chunk:
// |gelu'| <= 1.13, the dropout factor, |d act[t][f]| <= D max|W2| max|d ffout[t]|.
template <int D, int R, bool HS = false>
__global__ __launch_bounds__(D * 4) void xlnet_ff_bwd_kernel(FFBwdParams p) {
    constexpr int NW = D / 16, RT = 16 * R, PH = D + 16, PLN = RT * PH, DI = 4 * D;
    extern __shared__ uint16_t smem16[];
    uint16_t* sh_dfo = smem16;                                  // [3][RT][PH]  d ffout planes (B operand of the FF2 dX product)
    uint16_t* sh_dp = sh_dfo + 3 * PLN;                         // [3][RT][PH]  d pre chunk planes (B operand of the FF1 dX product)
    float* sh_part = reinterpret_cast<float*>(sh_dp + 3 * PLN);    // [NW][3][D]
    float* sh_inv_d = reinterpret_cast<float*>(sh_dfo + 2 * PLN);  // HS: [RT] inverse d ffout row scales (unused third plane)
    float* sh_sp = reinterpret_cast<float*>(sh_dp + 2 * PLN);      // HS: [RT] d pre scales | [RT] their inverses
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, g = lane >> 4;
    const long t0 = (long)blockIdx.x * RT;
    float dfomax_keep = 0.f;
    // ---- phase A: LayerNorm backward, lane = two consecutive features of a token row; a wave owns RT / NW rows and walks
    // them RB at a time with every load of the batch in flight before the first reduction (round 5: one row at a time was a
    // chain of RT / NW dependent load -> reduce -> store latencies, ~15 us of the launch at 10 rows per wave)
    {
        constexpr int RB = 5;
        const int c0 = lane * 2;
        const bool act_lane = c0 < D;
        float gam[2], pg[2], pb[2], pb2[2];
        float dfomax = 0.f;        // HS: running max |d ffout| over this wave's rows (operand maximum for the W2 weight gradient)
#pragma unroll
        for (int e = 0; e < 2; ++e) { gam[e] = act_lane ? p.gamma[c0 + e] : 0.f; pg[e] = pb[e] = pb2[e] = 0.f; }
        for (int row0 = w; row0 < RT; row0 += NW * RB) {
            float2 fo[RB], hh[RB], dd[RB];
            float mu_[RB], rs_[RB];
#pragma unroll
            for (int b = 0; b < RB; ++b) {          // unconditional loads from clamped addresses
                const long tc = min(t0 + min(row0 + b * NW, RT - 1), (long)p.T - 1);
                mu_[b] = p.mean[tc]; rs_[b] = p.rstd[tc];
                const int cc = act_lane ? c0 : 0;
                fo[b] = *reinterpret_cast<const float2*>(p.ffout + tc * D + cc);
                hh[b] = *reinterpret_cast<const float2*>(p.h1 + tc * D + cc);
                dd[b] = *reinterpret_cast<const float2*>(p.dy + tc * D + cc);
            }
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const int row = row0 + b * NW;
                if (row >= RT) break;           // wave-uniform
            const long t = t0 + row;
            float dxa[2] = {0.f, 0.f}, dx[2] = {0.f, 0.f};
            if (t < p.T) {      // wave-uniform
                const float mu = mu_[b], rs = rs_[b];
                float xh[2] = {0.f, 0.f}, gg[2] = {0.f, 0.f}, dyv[2] = {0.f, 0.f}, m[2] = {1.f, 1.f};
                float s1 = 0.f, s2 = 0.f;
                if (act_lane) {
                    drop_scale_vec<2>(p.drop_out, (unsigned long long)t * D + c0, true, m);
                    const float2 fo_ = fo[b], hh_ = hh[b], dd_ = dd[b];
                    const float xv[2] = {fo_.x * m[0] + hh_.x, fo_.y * m[1] + hh_.y};
                    dyv[0] = dd_.x; dyv[1] = dd_.y;
                    if (p.drop_final.p > 0.f) {       // workgroup-uniform
                        float mf[2];
                        drop_scale_vec<2>(p.drop_final, (unsigned long long)t * D + c0, true, mf);
                        dyv[0] *= mf[0]; dyv[1] *= mf[1];
                    }
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        xh[e] = (xv[e] - mu) * rs;
                        gg[e] = dyv[e] * gam[e];
                        s1 += gg[e];
                        s2 += gg[e] * xh[e];
                    }
                }
                s1 = wave_sum(s1) * (1.0f / D);
                s2 = wave_sum(s2) * (1.0f / D);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    dx[e] = act_lane ? rs * (gg[e] - s1 - xh[e] * s2) : 0.f;
                    dxa[e] = dx[e] * m[e];
                    pg[e] += dyv[e] * xh[e];
                    pb[e] += dyv[e];
                    pb2[e] += dxa[e];
                }
                if (act_lane) {
                    // residual part of d h1 (the FF1 dX product is added at the end) and the rows of the FF2 weight gradient
                    *reinterpret_cast<float2*>(p.dh1 + t * D + c0) = make_float2(dx[0], dx[1]);
                    *reinterpret_cast<float2*>(p.dffout + t * D + c0) = make_float2(dxa[0], dxa[1]);
                }
            }
            if constexpr (HS) {      // wave-uniform: the whole row lives in this wave
                float m = fmaxf(fabsf(dxa[0]), fabsf(dxa[1]));
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
                const float sc = pow2_scale(m);
                if (act_lane) {
                    uint32_t wd[2];
                    cut2h(dxa[0] * sc, dxa[1] * sc, wd);
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) *reinterpret_cast<uint32_t*>(sh_dfo + pl * PLN + row * PH + c0) = wd[pl];
                }
                dfomax = fmaxf(dfomax, m);
                if (lane == 0) {
                    sh_inv_d[row] = 1.f / sc;
                    const float w2max = 16384.f / p.wh.scale[HS_W2];       // >= max |W2|
                    const float sp = pow2_scale(1.13f * p.drop_act.inv_keep * (float)D * w2max * m);
                    sh_sp[row] = sp;
                    sh_sp[RT + row] = 1.f / sp;
                }
            } else if (act_lane) {
                uint32_t wd[3];
                cut3(dxa[0], dxa[1], wd);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint32_t*>(sh_dfo + pl * PLN + row * PH + c0) = wd[pl];
            }
            }
        }
        dfomax_keep = dfomax;
        if (act_lane) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                sh_part[(w * 3 + 0) * D + c0 + e] = pg[e];
                sh_part[(w * 3 + 1) * D + c0 + e] = pb[e];
                sh_part[(w * 3 + 2) * D + c0 + e] = pb2[e];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < 3 * D; i += NW * 64) {
        float s = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) s += sh_part[ww * 3 * D + i];
        p.partA[(long)blockIdx.x * 3 * D + i] = s;
    }
    // ---- phase B: d pre = (d ffout @ W2) * gelu'(pre) * mask, chunk by chunk; d h1 += d pre @ W1
    f32x4 acc3[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc3[r] = zero4();
    const int boff = n * PH + 8 * g;
    const long wpl = (long)DI * D;
    const uint16_t* w2tp = (HS ? p.wh.W2Tp : p.w.W2Tp) + (long)(16 * w + n) * D + 8 * g;     // + c D D: chunk c
    const uint16_t* w1tp = (HS ? p.wh.W1Tp : p.w.W1Tp) + (long)(16 * w + n) * DI + 8 * g;    // + c D
    AFrag<D> a2, a3;
    AFragH<D> h2f, h3f;
    float is2[R], is3[R], sps[R];      // HS: accumulator -> value factors of the two products, the d pre scales
    float dpremax = 0.f;               // HS: running max |d pre| of this lane (operand maximum for the W1 weight gradient)
    if constexpr (HS) {
        const float iw2 = 1.f / p.wh.scale[HS_W2], iw1 = 1.f / p.wh.scale[HS_W1];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            is2[r] = sh_inv_d[r * 16 + n] * iw2;
            sps[r] = sh_sp[r * 16 + n];
            is3[r] = sh_sp[RT + r * 16 + n] * iw1;
        }
        load_a2h<D>(h2f, w2tp, wpl);
    } else {
        load_a3<D>(a2, w2tp, wpl);
    }
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        const int d0 = c * D + 16 * w + 4 * g;
        if constexpr (HS) load_a2h<D>(h3f, w1tp + c * D, wpl);
        else load_a3<D>(a3, w1tp + c * D, wpl);
        float4 pre[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long t = min(t0 + r * 16 + n, (long)p.T - 1);
            pre[r] = ld4(p.ffpre + t * DI + d0);
        }
        f32x4 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = zero4();
        // d ff^T[d][token] = sum_f W2^T[d][f] d ffout[token][f]
        if constexpr (HS) {
            product3h<D, R, PH>(h2f, sh_dfo + boff, PLN, acc);
            load_a2h<D>(h2f, w2tp + (long)min(c + 1, 3) * D * D, wpl);
#pragma unroll
            for (int r = 0; r < R; ++r) { acc[r][0] *= is2[r]; acc[r][1] *= is2[r]; acc[r][2] *= is2[r]; acc[r][3] *= is2[r]; }
        } else {
            product3<D, R, PH>(a2, sh_dfo + boff, PLN, acc);
            load_a3<D>(a2, w2tp + (long)min(c + 1, 3) * D * D, wpl);
        }
        if (c > 0) __syncthreads();                           // the previous chunk's planes have been read by every wave
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long t = t0 + r * 16 + n;
            float4 v = make_float4(acc[r][0] * gelu_erf_grad(pre[r].x), acc[r][1] * gelu_erf_grad(pre[r].y),
                                   acc[r][2] * gelu_erf_grad(pre[r].z), acc[r][3] * gelu_erf_grad(pre[r].w));
            const float4 m = drop_scale4(p.drop_act, (unsigned long long)t * DI + d0);
            v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
            if (t < p.T) st4(p.dpre + t * DI + d0, v);           // rows beyond T are exact zeros (d ffout rows are)
            if constexpr (HS) {
                dpremax = fmaxf(dpremax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                uint32_t w0[2], w1[2];
                cut2h(v.x * sps[r], v.y * sps[r], w0);
                cut2h(v.z * sps[r], v.w * sps[r], w1);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    *reinterpret_cast<uint2*>(sh_dp + pl * PLN + (r * 16 + n) * PH + 16 * w + 4 * g) = make_uint2(w0[pl], w1[pl]);
            } else {
                uint32_t w0[3], w1[3];
                cut3(v.x, v.y, w0);
                cut3(v.z, v.w, w1);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    *reinterpret_cast<uint2*>(sh_dp + pl * PLN + (r * 16 + n) * PH + 16 * w + 4 * g) = make_uint2(w0[pl], w1[pl]);
            }
            cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
        }
        // d b1 partial: sum over the tile's tokens = over r (above) and over the 16 lanes n of a lane group
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            cs.x += __shfl_xor(cs.x, o, 64); cs.y += __shfl_xor(cs.y, o, 64);
            cs.z += __shfl_xor(cs.z, o, 64); cs.w += __shfl_xor(cs.w, o, 64);
        }
        if (n == 0) st4(p.partB + (long)blockIdx.x * DI + d0, cs);
        __syncthreads();
        // d h1^T[k][token] += sum_{d in chunk} W1^T[k][d] d pre[token][d]
        if constexpr (HS) product3h<D, R, PH>(h3f, sh_dp + boff, PLN, acc3);
        else product3<D, R, PH>(a3, sh_dp + boff, PLN, acc3);
    }
    if constexpr (HS) {
#pragma unroll
        for (int r = 0; r < R; ++r) { acc3[r][0] *= is3[r]; acc3[r][1] *= is3[r]; acc3[r][2] *= is3[r]; acc3[r][3] *= is3[r]; }
    }
    // d h1 = residual part (written in phase A by this workgroup, ordered by the barriers above) + d pre @ W1
    const int k0 = 16 * w + 4 * g;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long t = t0 + r * 16 + n;
        if (t < p.T) {
            const float4 o = ld4(p.dh1 + t * D + k0);
            st4(p.dh1 + t * D + k0, make_float4(o.x + acc3[r][0], o.y + acc3[r][1], o.z + acc3[r][2], o.w + acc3[r][3]));
        }
    }
    if constexpr (HS) {
        if (p.amax_dpre) {      // workgroup-uniform
            block_amax_to<NW>(dpremax, sh_part, p.amax_dpre, tid);
            block_amax_to<NW>(dfomax_keep, sh_part, p.amax_dfo, tid);
        }
    }
}

// -------------------------------------------------------------------------------------------------- host side
bool t4r_xlnet_body_fp16x2() {
    static int on = -1;
    if (on < 0) { const char* e = t4r_exp_getenv("T4R_XLNET_FP16X2"); on = e ? (atoi(e) != 0) : 1; }
    return on != 0;
}
// Where the next t4r_xlnet_ff_fwd / _bwd calls of this thread leave their per-workgroup operand maxima (four arrays of
// t4r_xlnet_ff_amax_slots(T) floats: max |h1|, max |act| from the forward, max |d pre|, max |d ffout| from the backward), or
// nulls (default: stand-alone use).  The layer (xlnet_layer.hip) sets them and hands the arrays to its two feed-forward
// weight-gradient launches, which then run in the two-way fp16 form (gemm_kernel.h: PREC 4).
static thread_local float* g_ff_amax[4] = {nullptr, nullptr, nullptr, nullptr};
// set and cleared inside one layer call (csrc/xlnet_layer.hip): the next t4r_xlnet_ff_fwd / _bwd of this thread also applies
// the model-level output dropout keyed by `ctr` (the last layer of a stack: one launch less per direction and no extra pass
// over [T, D]); 0 = off
static thread_local unsigned long long g_ff_final_ctr = 0;
static thread_local int g_ff_final_on = 0;
void t4r_xlnet_ff_final_dropout(int on, unsigned long long ctr) { g_ff_final_on = on; g_ff_final_ctr = ctr; }
void t4r_xlnet_ff_amax_buffers(float* h1, float* act, float* dpre, float* dfo) {
    g_ff_amax[0] = h1; g_ff_amax[1] = act; g_ff_amax[2] = dpre; g_ff_amax[3] = dfo;
}
static int pick_r(long T) { return t4r_xlnet_pick_r(T, false); }
int t4r_xlnet_ff_amax_count(long T) { const int R = pick_r(T); return (int)((T + 16 * R - 1) / (16 * R)); }   // workgroups of a forward launch
int t4r_xlnet_ff_amax_count_bwd(long T) { const int R = t4r_xlnet_pick_r(T, true); return (int)((T + 16 * R - 1) / (16 * R)); }   // of a backward launch
long t4r_xlnet_ff_amax_slots(long T) { return (T + 15) / 16; }                                                 // upper bound of it
// CUs the token-tile kernels of the BACKWARD pass may count on (0 = all 256).  Their default grid is ONE 512-thread workgroup
// per CU (256 tiles of 80 rows at 20 480 tokens): with k CUs held by something else -- an RCCL ring kernel reducing the table
// bucket under the body's backward, SURVEY 8(e) -- every such launch needs a second round of workgroups for k tiles and takes
// twice as long (tools/occupier_curve.py: 1.38x per step at k = 8 .. 64).  With a budget below 256 the backward launches
// take the tile that minimises rounds x rows-per-tile on that many CUs instead (48-row tiles at 240 CUs: two rounds of 0.6).
// Set by the data-parallel wiring while a collective is in flight (distributed.GradReducer); process-wide, read per launch.
static std::atomic<int> g_cu_budget{0};
extern "C" void t4r_xlnet_set_cu_budget(int cus) { g_cu_budget.store(cus < 0 ? 0 : cus, std::memory_order_relaxed); }
extern "C" int t4r_xlnet_get_cu_budget(void) { return g_cu_budget.load(std::memory_order_relaxed); }
// compute units of the current device (256 on MI355X), asked once per device
static int device_cus() {
    static std::atomic<int> cus[T4R_MAX_DEVICES] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::atomic<int>& c = cus[dev & (T4R_MAX_DEVICES - 1)];
    int n = c.load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        c.store(n, std::memory_order_relaxed);
    }
    return n;
}
extern "C" int t4r_device_cus(void) { return device_cus(); }
int t4r_xlnet_pick_r(long T, bool backward) {
    const long blocks16 = (T + 15) / 16;
    const int ncu = device_cus();
    const int budget = backward ? g_cu_budget.load(std::memory_order_relaxed) : 0;
    if (budget <= 0 || budget >= ncu) {
        // smallest tile (fewest padded rows) whose grid still fits one residency of the chip's CUs; 5 beyond that
        if (blocks16 <= ncu) return 1;
        if (blocks16 <= 2L * ncu) return 2;
        if (blocks16 <= 3L * ncu) return 3;
        return 5;
    }
    int best = 5;
    long best_cost = -1;
    const int cand[4] = {5, 3, 2, 1};
    for (int i = 0; i < 4; ++i) {
        const int R = cand[i];
        const long tiles = (blocks16 + R - 1) / R, rounds = (tiles + budget - 1) / budget;
        const long cost = rounds * R;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = R; }     // ties: the larger tile (fewer weight streams)
    }
    return best;
}

extern "C" int t4r_xlnet_fused_supported(int D) { return D == 32 || D == 64 || D == 128; }
// matrix instructions per fp32-equivalent one in the fused layer kernels: 3 (two-way fp16 split, default) or 6 (bf16 planes)
extern "C" int t4r_xlnet_fused_products(void) { return t4r_xlnet_body_fp16x2() ? 3 : 6; }
extern "C" long t4r_xlnet_ff_bwd_part_floats(long T, int D) { return ((T + 15) / 16) * 7L * D; }
template <int D, int R>
static int ff_fwd_launch(hipStream_t st, const FFFwdParams& p) {
    constexpr int RT = 16 * R, PH = D + 16, NW = D / 16;
    const size_t smem = (size_t)(6 * RT * PH) * 2 + (size_t)(2 * NW * RT) * sizeof(float);
    const dim3 grid((unsigned)((p.T + RT - 1) / RT));
    if (t4r_xlnet_body_fp16x2()) {
        static T4rLdsAttr attr;
        t4r_ensure_dynamic_lds((const void*)xlnet_ff_fwd_kernel<D, R, true>, smem, attr);
        hipLaunchKernelGGL((xlnet_ff_fwd_kernel<D, R, true>), grid, dim3(D * 4), smem, st, p);
    } else {
        static T4rLdsAttr attr;
        t4r_ensure_dynamic_lds((const void*)xlnet_ff_fwd_kernel<D, R>, smem, attr);
        hipLaunchKernelGGL((xlnet_ff_fwd_kernel<D, R>), grid, dim3(D * 4), smem, st, p);
    }
    T4R_LAUNCH_CHECK();
    return 0;
}
template <int D, int R>
static int ff_bwd_launch(hipStream_t st, const FFBwdParams& p) {
    constexpr int RT = 16 * R, PH = D + 16, NW = D / 16;
    const size_t smem = (size_t)(6 * RT * PH) * 2 + (size_t)(NW * 3 * D) * sizeof(float);
    const dim3 grid((unsigned)((p.T + RT - 1) / RT));
    if (t4r_xlnet_body_fp16x2()) {
        static T4rLdsAttr attr;
        t4r_ensure_dynamic_lds((const void*)xlnet_ff_bwd_kernel<D, R, true>, smem, attr);
        hipLaunchKernelGGL((xlnet_ff_bwd_kernel<D, R, true>), grid, dim3(D * 4), smem, st, p);
    } else {
        static T4rLdsAttr attr;
        t4r_ensure_dynamic_lds((const void*)xlnet_ff_bwd_kernel<D, R>, smem, attr);
        hipLaunchKernelGGL((xlnet_ff_bwd_kernel<D, R>), grid, dim3(D * 4), smem, st, p);
    }
    T4R_LAUNCH_CHECK();
    return 0;
}

#define FUSED_DISPATCH(FN, D, R, ...)                                                           \
    do {                                                                                        \
        switch ((D) * 8 + (R)) {                                                                \
            case 32 * 8 + 1: return FN<32, 1>(__VA_ARGS__);   case 32 * 8 + 2: return FN<32, 2>(__VA_ARGS__);   \
            case 32 * 8 + 3: return FN<32, 3>(__VA_ARGS__);   case 32 * 8 + 5: return FN<32, 5>(__VA_ARGS__);   \
            case 64 * 8 + 1: return FN<64, 1>(__VA_ARGS__);   case 64 * 8 + 2: return FN<64, 2>(__VA_ARGS__);   \
            case 64 * 8 + 3: return FN<64, 3>(__VA_ARGS__);   case 64 * 8 + 5: return FN<64, 5>(__VA_ARGS__);   \
            case 128 * 8 + 1: return FN<128, 1>(__VA_ARGS__); case 128 * 8 + 2: return FN<128, 2>(__VA_ARGS__); \
            case 128 * 8 + 3: return FN<128, 3>(__VA_ARGS__); case 128 * 8 + 5: return FN<128, 5>(__VA_ARGS__); \
        }                                                                                       \
    } while (0)

// Fused feed-forward block, forward.  h1 [T, D] -> hout [T, D]; saves ffpre / ffact [T, 4D], ffout [T, D], mean / rstd [T]
// for the backward when the pointers are given (all or none).  planes: the buffer t4r_xlnet_ff_prepare filled.
// replaces: HF modeling_xlnet.py:297-305 (layer_1, activation, dropout, layer_2, dropout, layer_norm(output + inp))
extern "C" int t4r_xlnet_ff_fwd(void* stream, const float* h1, const float* planes, const float* b1, const float* b2,
                                const float* gamma, const float* beta, float* ffpre, float* ffact, float* ffout, float* mean,
                                float* rstd, float* hout, int T, int D, float eps, float drop_p, unsigned long long seed,
                                unsigned long long ctr_act, unsigned long long ctr_out) {
    if (T <= 0) return 0;
    T4R_CHECK_ARG(t4r_xlnet_fused_supported(D), "xlnet_ff_fwd: d_model must be 32, 64 or 128");
    T4R_CHECK_ARG(h1 && planes && b1 && b2 && gamma && beta && hout, "xlnet_ff_fwd: null pointer");
    const bool train = ffpre != nullptr;
    T4R_CHECK_ARG((ffact != nullptr) == train && (ffout != nullptr) == train && (mean != nullptr) == train &&
                      (rstd != nullptr) == train, "xlnet_ff_fwd: ffpre, ffact, ffout, mean, rstd are saved together or not at all");
    T4R_CHECK_ARG(train || drop_p == 0.f, "xlnet_ff_fwd: dropout needs the saved activations (training mode)");
    FFFwdParams p{h1, b1, b2, gamma, beta, carve_planes(planes, D), carve_planes_h(planes, D), g_ff_amax[0], g_ff_amax[1], ffpre, ffact, ffout,
                  mean, rstd, hout, T, eps,
                  make_drop(drop_p, seed, ctr_act), make_drop(drop_p, seed, ctr_out),
                  make_drop(g_ff_final_on ? drop_p : 0.f, seed, g_ff_final_ctr)};
    const int R = pick_r(T);
    FUSED_DISPATCH(ff_fwd_launch, D, R, (hipStream_t)stream, p);
    t4r_set_error("xlnet_ff_fwd: no instantiation");
    return -1;
}

// Fused feed-forward block, backward (input gradients + bias / LayerNorm parameter gradients; the two weight gradients
// d W2 += d ffout^T @ ffact and d W1 += d pre^T @ h1 are token-reducing GEMMs the caller issues on the rows written here).
//   dy [T, D] = d loss / d hout;  dh1 [T, D] (overwritten) = d loss / d h1;  dffout [T, D], dpre [T, 4D] (overwritten);
//   d_gamma, d_beta, d_b2 [D], d_b1 [4D] ACCUMULATED;  part: t4r_xlnet_ff_bwd_part_floats(T, D) floats of scratch.
extern "C" int t4r_xlnet_ff_bwd(void* stream, const float* dy, const float* ffout, const float* h1, const float* mean,
                                const float* rstd, const float* gamma, const float* ffpre, const float* planes,
                                float* dh1, float* dffout, float* dpre, float* d_gamma, float* d_beta,
                                float* d_b2, float* d_b1, float* part, int T, int D, float drop_p,
                                unsigned long long seed, unsigned long long ctr_act, unsigned long long ctr_out) {
    if (T <= 0) return 0;
    T4R_CHECK_ARG(t4r_xlnet_fused_supported(D), "xlnet_ff_bwd: d_model must be 32, 64 or 128");
    T4R_CHECK_ARG(dy && ffout && h1 && mean && rstd && gamma && ffpre && planes && dh1 && dffout && dpre && part,
                  "xlnet_ff_bwd: null pointer");
    const int R = t4r_xlnet_pick_r(T, true);
    const int nwg = (T + 16 * R - 1) / (16 * R);
    float* partA = part;
    float* partB = part + (long)nwg * 3 * D;
    FFBwdParams p{dy, ffout, h1, mean, rstd, gamma, ffpre, carve_planes(planes, D), carve_planes_h(planes, D), g_ff_amax[2], g_ff_amax[3], dh1,
                  dffout, dpre, partA, partB, T,
                  make_drop(drop_p, seed, ctr_act), make_drop(drop_p, seed, ctr_out),
                  make_drop(g_ff_final_on ? drop_p : 0.f, seed, g_ff_final_ctr)};
    hipStream_t st = (hipStream_t)stream;
    auto launch = [&]() -> int {
        FUSED_DISPATCH(ff_bwd_launch, D, R, st, p);
        t4r_set_error("xlnet_ff_bwd: no instantiation");
        return -1;
    };
    const int rc = launch();
    if (rc != 0) return rc;
    int rc2 = t4r_reduce_partials_launch(st, partA, nwg, d_gamma, D, 1, d_beta, D, 1, d_b2, D, 1);
    if (rc2 != 0) return rc2;
    return t4r_reduce_partials_launch(st, partB, nwg, d_b1, 4 * D, 1, nullptr, 0, 0, nullptr, 0, 0);
}
