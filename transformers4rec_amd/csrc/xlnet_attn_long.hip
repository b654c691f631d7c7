// XLNet relative-position attention core for sequences LONGER than one wave (L > 64) and for head widths the one-wave kernels
// have no instance for (d_head up to 256 other than 8 / 16 / 32), forward and backward (gfx950).
//
// The reference takes any total_seq_length (XLNetConfig.build, transformers4rec/config/transformer.py:432-482; HF
// modeling_xlnet.py rel_attn_core :95-140, rel_shift_bnij :81-93, reached through transformers4rec/torch/block/transformer.py:
// 179-199); the kernels of xlnet_attn.hip / xlnet_attn_mfma.hip / xlnet_attn_block.hip put one query row on one lane of ONE
// wave and stop at 64 (32) positions.  This file is the general form behind the same entry points (t4r_xlnet_attn_fwd / _bwd pick
// it when L > 64, when d_head is not 8 / 16 / 32, or when a session's rows overflow the LDS of the one-wave VALU kernels; the
// second half of the file is the same for t4r_mha_fwd / _bwd of the GPT-2 / BERT blocks): same arithmetic, same dropout keys (mask index ((b n_head + h) L + i) L + j), same opt-in key mask, no bound
// on L other than memory.  It is the coverage path, not the benchmarked one: everything is recomputed from q, k, v, k_r with
// plain fp32 FMAs, rows of k / v / q / d out come as wave-uniform (broadcast) loads out of L2, the lane-dependent row of the
// other operand as a 16-byte-per-lane gather (element by element when d_head is not a multiple of 4); no LDS, so no L x d_model
// limit either.  Head widths up to 256: the per-thread vectors have a compile-time capacity (8 .. 256) and a run-time width.
//
//   forward          thread = query row i (blocks of 64 rows walked by one wave per (session, head)); one online-softmax pass
//                    over the keys; saves the row log-sum-exp
//   backward, rows   thread = query row i:  delta_i = d out_i . out_i (saved for the other two passes),
//                    dS_ij = scale P_ij (mask_ij d out_i . v_j - delta_i),  d q_i = sum_j dS_ij (k_j + k_r[L + j - i]);
//                    the two halves of d q summed over rows and sessions are d r_w_bias / d r_r_bias (per-workgroup partials)
//   backward, keys   thread = key j:  d k_j = sum_i dS_ij (q_i + r_w_bias),  d v_j = sum_i P~_ij d out_i
//   backward, rel    thread = relative position m:  d k_r[m] = sum_{i, j = m - L + i} dS_ij (q_i + r_r_bias)   (per session, or
//                    summed over the sessions of the workgroup for the shared k_r: partials reduced in block order)
// Batch-reduced gradients go through the same partial buffer and reduction launch as the short kernels: deterministic.
#include "t4r_common.h"
#include <stdlib.h>

#define T4R_KEY_MASKED (-1e30f)

namespace {

// four consecutive floats of a head's row: one 16-byte request when the head width is a multiple of 4 (v4: rows and head offsets
// are then 16-byte aligned), else element by element with the row's tail (left elements) zero-filled
__device__ __forceinline__ float4 ld4g(const float* __restrict__ p, int left, bool v4) {
    if (v4) return *reinterpret_cast<const float4*>(p);
    return make_float4(p[0], left > 1 ? p[1] : 0.f, left > 2 ? p[2] : 0.f, left > 3 ? p[3] : 0.f);
}
__device__ __forceinline__ void st4g(float* __restrict__ p, float4 v, int left, bool v4) {
    if (v4) { *reinterpret_cast<float4*>(p) = v; return; }
    p[0] = v.x;
    if (left > 1) p[1] = v.y;
    if (left > 2) p[2] = v.z;
    if (left > 3) p[3] = v.w;
}
// DH is the CAPACITY of the per-thread vectors (registers: every index is a compile-time constant); dh <= DH, a multiple of 4, is
// the head width of the call -- the guard per 16-byte group is a scalar compare
template <int DH>
__device__ __forceinline__ void load_row(float (&x)[DH], const float* __restrict__ p, int dh, bool v4) {
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
        if (d < dh) {
            const float4 t = ld4g(p + d, dh - d, v4);
            x[d] = t.x; x[d + 1] = t.y; x[d + 2] = t.z; x[d + 3] = t.w;
        } else {
            x[d] = 0.f; x[d + 1] = 0.f; x[d + 2] = 0.f; x[d + 3] = 0.f;
        }
    }
}
template <int DH>
__device__ __forceinline__ float dot_row(const float (&a)[DH], const float* __restrict__ p, int dh, bool v4) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
        if (d < dh) {
            const float4 t = ld4g(p + d, dh - d, v4);
            s += a[d] * t.x + a[d + 1] * t.y + a[d + 2] * t.z + a[d + 3] * t.w;
        }
    }
    return s;
}
// (a + bias) . p
template <int DH>
__device__ __forceinline__ float dot_row_bias(const float* __restrict__ a, const float* __restrict__ bias, const float (&x)[DH], int dh, bool v4) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
        if (d < dh) {
            const float4 t = ld4g(a + d, dh - d, v4);
            const float4 u = ld4g(bias + d, dh - d, v4);
            s += (t.x + u.x) * x[d] + (t.y + u.y) * x[d + 1] + (t.z + u.z) * x[d + 2] + (t.w + u.w) * x[d + 3];
        }
    }
    return s;
}
template <int DH>
__device__ __forceinline__ void axpy_row(float (&acc)[DH], float a, const float* __restrict__ p, int dh, bool v4) {
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
        if (d < dh) {
            const float4 t = ld4g(p + d, dh - d, v4);
            acc[d] += a * t.x; acc[d + 1] += a * t.y; acc[d + 2] += a * t.z; acc[d + 3] += a * t.w;
        }
    }
}
template <int DH>
__device__ __forceinline__ void axpy_row_bias(float (&acc)[DH], float a, const float* __restrict__ p, const float* __restrict__ bias, int dh, bool v4) {
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
        if (d < dh) {
            const float4 t = ld4g(p + d, dh - d, v4);
            const float4 u = ld4g(bias + d, dh - d, v4);
            acc[d] += a * (t.x + u.x); acc[d + 1] += a * (t.y + u.y); acc[d + 2] += a * (t.z + u.z); acc[d + 3] += a * (t.w + u.w);
        }
    }
}

struct LongArgs {
    const float *q, *k, *v, *kr, *rw, *rr;     // [B L, D] x 3; k_r [2 L, D] (+ b kr_bstride); biases [D]
    const float *out, *lse, *dout;             // backward: forward output, row log-sum-exp [B, n, L], upstream gradient
    float *o, *lse_o;                          // forward outputs
    float *dq, *dk, *dv, *dkr_b, *part, *delta;
    int B, L, n_head, dh;
    float scale;
    long kr_bstride;
    DropCfg drop;
    const int* key_len;
};

// score of (i, j) for the thread that holds (q_i + r_w_bias) and (q_i + r_r_bias)
#define T4R_MASKED(j, i, klen) ((j) >= (klen) && (j) != (i))

template <int DH>
__global__ __launch_bounds__(64) void attn_long_fwd_kernel(LongArgs a) {
    const int L = a.L, dh = a.dh, D = a.n_head * dh, h = blockIdx.y, hc = h * dh, lane = threadIdx.x;
    const bool v4 = (dh & 3) == 0;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float* krb = a.kr + (long)b * a.kr_bstride + hc;
        const int klen = a.key_len ? a.key_len[b] : L;
        for (int i0 = 0; i0 < L; i0 += 64) {
            const int i = i0 + lane, ic = min(i, L - 1);
            float qw[DH], qr[DH], o[DH];
            {
                const float* qrow = a.q + ((long)b * L + ic) * D + hc;
#pragma unroll
                for (int d = 0; d < DH; ++d) {
                    const int dc = d < dh ? d : 0;
                    const float t = qrow[dc];
                    qw[d] = d < dh ? t + a.rw[hc + dc] : 0.f; qr[d] = d < dh ? t + a.rr[hc + dc] : 0.f; o[d] = 0.f;
                }
            }
            float m = -INFINITY, l = 0.f;
            for (int j = 0; j < L; ++j) {
                float s = dot_row<DH>(qw, a.k + ((long)b * L + j) * D + hc, dh, v4) + dot_row<DH>(qr, krb + (long)(j + L - ic) * D, dh, v4);
                s *= a.scale;
                if (T4R_MASKED(j, ic, klen)) s = T4R_KEY_MASKED;
                const float mn = fmaxf(m, s);
                const float alpha = __expf(m - mn), pj = __expf(s - mn);
                l = l * alpha + pj;
                float pd = pj;
                if (a.drop.p > 0.f) pd *= drop_scale(a.drop, ((unsigned long long)(b * a.n_head + h) * L + ic) * L + j);
#pragma unroll
                for (int d = 0; d < DH; ++d) o[d] *= alpha;
                axpy_row<DH>(o, pd, a.v + ((long)b * L + j) * D + hc, dh, v4);
                m = mn;
            }
            if (i < L) {
                const float inv = 1.f / l;
                float* orow = a.o + ((long)b * L + i) * D + hc;
#pragma unroll
                for (int d = 0; d < DH; d += 4)
                    if (d < dh) st4g(orow + d, make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv), dh - d, v4);
                a.lse_o[((long)b * a.n_head + h) * L + i] = m + __logf(l);
            }
        }
    }
}

// dS_ij / scale-free pieces shared by the three backward passes
__device__ __forceinline__ float prob_of(float s, float lse) { return __expf(s - lse); }

template <int DH>
__global__ __launch_bounds__(64) void attn_long_bwd_rows_kernel(LongArgs a) {
    const int L = a.L, dh = a.dh, D = a.n_head * dh, h = blockIdx.y, hc = h * dh, lane = threadIdx.x;
    const bool v4 = (dh & 3) == 0;
    float srw[DH], srr[DH];                     // this lane's share of d r_w_bias / d r_r_bias (all its rows and sessions)
#pragma unroll
    for (int d = 0; d < DH; ++d) { srw[d] = 0.f; srr[d] = 0.f; }
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float* krb = a.kr + (long)b * a.kr_bstride + hc;
        const int klen = a.key_len ? a.key_len[b] : L;
        for (int i0 = 0; i0 < L; i0 += 64) {
            const int i = i0 + lane, ic = min(i, L - 1);
            const bool live = i < L;
            float qw[DH], qr[DH], g[DH], dqa[DH], dqb[DH];
            const long row = ((long)b * L + ic) * D + hc;
            float delta = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) {
                const bool on = d < dh;
                const int dc = on ? d : 0;
                const float t = a.q[row + dc];
                qw[d] = on ? t + a.rw[hc + dc] : 0.f; qr[d] = on ? t + a.rr[hc + dc] : 0.f;
                g[d] = on ? a.dout[row + dc] : 0.f;
                delta += g[d] * a.out[row + dc];
                dqa[d] = 0.f; dqb[d] = 0.f;
            }
            const float lrow = a.lse[((long)b * a.n_head + h) * L + ic];
            if (live) a.delta[((long)b * a.n_head + h) * L + i] = delta;
            for (int j = 0; j < L; ++j) {
                const float* kj = a.k + ((long)b * L + j) * D + hc;
                const float* krm = krb + (long)(j + L - ic) * D;
                float s = (dot_row<DH>(qw, kj, dh, v4) + dot_row<DH>(qr, krm, dh, v4)) * a.scale;
                if (T4R_MASKED(j, ic, klen)) s = T4R_KEY_MASKED;
                const float p = prob_of(s, lrow);
                float dp = dot_row<DH>(g, a.v + ((long)b * L + j) * D + hc, dh, v4);
                if (a.drop.p > 0.f) dp *= drop_scale(a.drop, ((unsigned long long)(b * a.n_head + h) * L + ic) * L + j);
                const float ds = p * (dp - delta) * a.scale;
                axpy_row<DH>(dqa, ds, kj, dh, v4);
                axpy_row<DH>(dqb, ds, krm, dh, v4);
            }
            if (live) {
                float* dqrow = a.dq + ((long)b * L + i) * D + hc;
#pragma unroll
                for (int d = 0; d < DH; d += 4)
                    if (d < dh) st4g(dqrow + d, make_float4(dqa[d] + dqb[d], dqa[d + 1] + dqb[d + 1], dqa[d + 2] + dqb[d + 2], dqa[d + 3] + dqb[d + 3]), dh - d, v4);
#pragma unroll
                for (int d = 0; d < DH; ++d) { srw[d] += dqa[d]; srr[d] += dqb[d]; }
            }
        }
    }
    // the workgroup's (= the wave's) sums over its lanes, lane order fixed by the butterfly
    float* mypart = a.part + (long)blockIdx.x * (2L * L * D + 2 * D) + 2L * L * D;
#pragma unroll
    for (int d = 0; d < DH; ++d) {
        float x = srw[d], y = srr[d];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { x += __shfl_xor(x, o, 64); y += __shfl_xor(y, o, 64); }
        if (lane == 0 && d < dh) { mypart[hc + d] = x; mypart[D + hc + d] = y; }
    }
}

template <int DH>
__global__ __launch_bounds__(64) void attn_long_bwd_keys_kernel(LongArgs a) {
    const int L = a.L, dh = a.dh, D = a.n_head * dh, h = blockIdx.y, hc = h * dh, lane = threadIdx.x;
    const bool v4 = (dh & 3) == 0;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float* krb = a.kr + (long)b * a.kr_bstride + hc;
        const int klen = a.key_len ? a.key_len[b] : L;
        for (int j0 = 0; j0 < L; j0 += 64) {
            const int j = j0 + lane, jc = min(j, L - 1);
            float kj[DH], vj[DH], dk[DH], dv[DH];
            const long row = ((long)b * L + jc) * D + hc;
            load_row<DH>(kj, a.k + row, dh, v4);
            load_row<DH>(vj, a.v + row, dh, v4);
#pragma unroll
            for (int d = 0; d < DH; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
            for (int i = 0; i < L; ++i) {
                const float* qi = a.q + ((long)b * L + i) * D + hc;          // wave-uniform rows
                const float* gi = a.dout + ((long)b * L + i) * D + hc;
                float krm[DH];
                load_row<DH>(krm, krb + (long)(jc + L - i) * D, dh, v4);
                float s = (dot_row_bias<DH>(qi, a.rw + hc, kj, dh, v4) + dot_row_bias<DH>(qi, a.rr + hc, krm, dh, v4)) * a.scale;
                if (T4R_MASKED(jc, i, klen)) s = T4R_KEY_MASKED;
                const float p = prob_of(s, a.lse[((long)b * a.n_head + h) * L + i]);
                float ms = 1.f;
                if (a.drop.p > 0.f) ms = drop_scale(a.drop, ((unsigned long long)(b * a.n_head + h) * L + i) * L + jc);
                const float dp = dot_row<DH>(vj, gi, dh, v4) * ms;
                const float ds = p * (dp - a.delta[((long)b * a.n_head + h) * L + i]) * a.scale;
                axpy_row_bias<DH>(dk, ds, qi, a.rw + hc, dh, v4);
                axpy_row<DH>(dv, p * ms, gi, dh, v4);
            }
            if (j < L) {
#pragma unroll
                for (int d = 0; d < DH; d += 4) {
                    if (d < dh) {
                        st4g(a.dk + row + d, make_float4(dk[d], dk[d + 1], dk[d + 2], dk[d + 3]), dh - d, v4);
                        st4g(a.dv + row + d, make_float4(dv[d], dv[d + 1], dv[d + 2], dv[d + 3]), dh - d, v4);
                    }
                }
            }
        }
    }
}

template <int DH, bool SHARED_KR>
__global__ __launch_bounds__(64) void attn_long_bwd_rel_kernel(LongArgs a) {
    const int L = a.L, dh = a.dh, D = a.n_head * dh, h = blockIdx.y, hc = h * dh, lane = threadIdx.x;
    const bool v4 = (dh & 3) == 0;
    for (int m0 = 0; m0 < 2 * L; m0 += 64) {
        const int m = m0 + lane, mc = min(m, 2 * L - 1);
        float acc[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) acc[d] = 0.f;
        for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
            const float* krb = a.kr + (long)b * a.kr_bstride + hc;
            const int klen = a.key_len ? a.key_len[b] : L;
            float krm[DH];
            load_row<DH>(krm, krb + (long)mc * D, dh, v4);
            if (!SHARED_KR) {
#pragma unroll
                for (int d = 0; d < DH; ++d) acc[d] = 0.f;
            }
            for (int i = 0; i < L; ++i) {
                const int j = mc - L + i;                       // the key this relative position meets query i at
                const bool hit = j >= 0 && j < L;
                const int jc = min(max(j, 0), L - 1);
                const float* qi = a.q + ((long)b * L + i) * D + hc;
                const float* gi = a.dout + ((long)b * L + i) * D + hc;
                float kj[DH];
                load_row<DH>(kj, a.k + ((long)b * L + jc) * D + hc, dh, v4);
                float s = (dot_row_bias<DH>(qi, a.rw + hc, kj, dh, v4) + dot_row_bias<DH>(qi, a.rr + hc, krm, dh, v4)) * a.scale;
                if (T4R_MASKED(jc, i, klen)) s = T4R_KEY_MASKED;
                const float p = hit ? prob_of(s, a.lse[((long)b * a.n_head + h) * L + i]) : 0.f;
                float dp;
                {
                    float vj[DH];
                    load_row<DH>(vj, a.v + ((long)b * L + jc) * D + hc, dh, v4);
                    dp = dot_row<DH>(vj, gi, dh, v4);
                }
                if (a.drop.p > 0.f) dp *= drop_scale(a.drop, ((unsigned long long)(b * a.n_head + h) * L + i) * L + jc);
                const float ds = p * (dp - a.delta[((long)b * a.n_head + h) * L + i]) * a.scale;
                axpy_row_bias<DH>(acc, ds, qi, a.rr + hc, dh, v4);
            }
            if (!SHARED_KR && m < 2 * L) {
                float* o = a.dkr_b + ((long)b * 2 * L + m) * D + hc;
#pragma unroll
                for (int d = 0; d < DH; d += 4)
                    if (d < dh) st4g(o + d, make_float4(acc[d], acc[d + 1], acc[d + 2], acc[d + 3]), dh - d, v4);
            }
        }
        if (SHARED_KR && m < 2 * L) {
            float* o = a.part + (long)blockIdx.x * (2L * L * D + 2 * D) + (long)m * D + hc;
#pragma unroll
            for (int d = 0; d < DH; d += 4)
                if (d < dh) st4g(o + d, make_float4(acc[d], acc[d + 1], acc[d + 2], acc[d + 3]), dh - d, v4);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same for the scaled-dot-product core of the GPT-2 / BERT blocks (mha.hip: one workgroup per (session, head) with the
// head's K / V in LDS stops at 128 positions and d_head 16 / 32 / 64; HF gpt2/modeling_gpt2.py eager_attention_forward :54-72,
// bert/modeling_bert.py BertSelfAttention, built by transformers4rec/config/transformer.py:218-260, :493-534 for ANY
// total_seq_length / d_model).  Same conventions as mha.hip: keys j < min(causal ? i + 1 : L, clamp(key_len[b], 1, L)) take
// part; dropout mask index ((b n_head + h) L + i) L + j; rows of ld / ld_out / ld_d floats.
struct MhaLongArgs {
    const float *q, *k, *v, *out, *dout, *lse;
    float *o, *lse_o, *dq, *dk, *dv;
    long ld, ld_out, ld_d;
    int B, L, n_head, dh, causal;
    float scale;
    DropCfg drop;
    const int* key_len;
};

template <int DH>
__global__ __launch_bounds__(64) void mha_long_fwd_kernel(MhaLongArgs a) {
    const int L = a.L, dh = a.dh, h = blockIdx.y, hc = h * dh, lane = threadIdx.x;
    const bool v4 = (dh & 3) == 0;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const int klen = a.key_len ? max(1, min(L, a.key_len[b])) : L;
        for (int i0 = 0; i0 < L; i0 += 64) {
            const int i = i0 + lane, ic = min(i, L - 1);
            float qi[DH], o[DH];
            load_row<DH>(qi, a.q + ((long)b * L + ic) * a.ld + hc, dh, v4);
#pragma unroll
            for (int d = 0; d < DH; ++d) o[d] = 0.f;
            float m = -INFINITY, l = 0.f;
            const int jend = min(a.causal ? ic + 1 : L, klen);
            const int jmax = min(a.causal ? min(i0 + 64, L) : L, klen);       // wave-uniform bound: the rows of k / v stay broadcasts
            for (int j = 0; j < jmax; ++j) {
                const bool on = j < jend;
                const float s = on ? dot_row<DH>(qi, a.k + ((long)b * L + j) * a.ld + hc, dh, v4) * a.scale : -INFINITY;
                const float mn = fmaxf(m, s);
                const float alpha = mn == -INFINITY ? 1.f : __expf(m - mn), pj = on ? __expf(s - mn) : 0.f;
                l = l * alpha + pj;
                float pd = pj;
                if (a.drop.p > 0.f) pd *= drop_scale(a.drop, ((unsigned long long)(b * a.n_head + h) * L + ic) * L + j);
#pragma unroll
                for (int d = 0; d < DH; ++d) o[d] *= alpha;
                axpy_row<DH>(o, pd, a.v + ((long)b * L + j) * a.ld + hc, dh, v4);
                m = mn;
            }
            if (i < L) {
                const float inv = 1.f / l;
                float* orow = a.o + ((long)b * L + i) * a.ld_out + hc;
#pragma unroll
                for (int d = 0; d < DH; d += 4)
                    if (d < dh) st4g(orow + d, make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv), dh - d, v4);
                a.lse_o[((long)b * a.n_head + h) * L + i] = m + __logf(l);
            }
        }
    }
}

// d q: thread = query row
template <int DH>
__global__ __launch_bounds__(64) void mha_long_bwd_rows_kernel(MhaLongArgs a) {
    const int L = a.L, dh = a.dh, h = blockIdx.y, hc = h * dh, lane = threadIdx.x;
    const bool v4 = (dh & 3) == 0;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const int klen = a.key_len ? max(1, min(L, a.key_len[b])) : L;
        for (int i0 = 0; i0 < L; i0 += 64) {
            const int i = i0 + lane, ic = min(i, L - 1);
            float qi[DH], g[DH], dq[DH];
            load_row<DH>(qi, a.q + ((long)b * L + ic) * a.ld + hc, dh, v4);
            load_row<DH>(g, a.dout + ((long)b * L + ic) * a.ld_out + hc, dh, v4);
            const float delta = dot_row<DH>(g, a.out + ((long)b * L + ic) * a.ld_out + hc, dh, v4);
#pragma unroll
            for (int d = 0; d < DH; ++d) dq[d] = 0.f;
            const float lrow = a.lse[((long)b * a.n_head + h) * L + ic];
            const int jend = min(a.causal ? ic + 1 : L, klen);
            const int jmax = min(a.causal ? min(i0 + 64, L) : L, klen);
            for (int j = 0; j < jmax; ++j) {
                const float* kj = a.k + ((long)b * L + j) * a.ld + hc;
                const float p = j < jend ? __expf(dot_row<DH>(qi, kj, dh, v4) * a.scale - lrow) : 0.f;
                float dp = dot_row<DH>(g, a.v + ((long)b * L + j) * a.ld + hc, dh, v4);
                if (a.drop.p > 0.f) dp *= drop_scale(a.drop, ((unsigned long long)(b * a.n_head + h) * L + ic) * L + j);
                axpy_row<DH>(dq, p * (dp - delta) * a.scale, kj, dh, v4);
            }
            if (i < L) {
                float* o = a.dq + ((long)b * L + i) * a.ld_d + hc;
#pragma unroll
                for (int d = 0; d < DH; d += 4)
                    if (d < dh) st4g(o + d, make_float4(dq[d], dq[d + 1], dq[d + 2], dq[d + 3]), dh - d, v4);
            }
        }
    }
}

// d k, d v: thread = key; the rows of q / d out / out are wave-uniform, delta_i = d out_i . out_i is recomputed per row
template <int DH>
__global__ __launch_bounds__(64) void mha_long_bwd_keys_kernel(MhaLongArgs a) {
    const int L = a.L, dh = a.dh, h = blockIdx.y, hc = h * dh, lane = threadIdx.x;
    const bool v4 = (dh & 3) == 0;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const int klen = a.key_len ? max(1, min(L, a.key_len[b])) : L;
        for (int j0 = 0; j0 < L; j0 += 64) {
            const int j = j0 + lane, jc = min(j, L - 1);
            float kj[DH], vj[DH], dk[DH], dv[DH];
            load_row<DH>(kj, a.k + ((long)b * L + jc) * a.ld + hc, dh, v4);
            load_row<DH>(vj, a.v + ((long)b * L + jc) * a.ld + hc, dh, v4);
#pragma unroll
            for (int d = 0; d < DH; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
            const bool key_on = jc < klen;
            for (int i = a.causal ? j0 : 0; i < L; ++i) {
                const float* qi = a.q + ((long)b * L + i) * a.ld + hc;
                const float* gi = a.dout + ((long)b * L + i) * a.ld_out + hc;
                const float* oi = a.out + ((long)b * L + i) * a.ld_out + hc;
                float delta = 0.f;
#pragma unroll
                for (int d = 0; d < DH; d += 4) {
                    if (d < dh) {
                        const float4 x = ld4g(gi + d, dh - d, v4), y = ld4g(oi + d, dh - d, v4);
                        delta += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
                    }
                }
                const bool on = key_on && (!a.causal || jc <= i);
                const float p = on ? __expf(dot_row<DH>(kj, qi, dh, v4) * a.scale - a.lse[((long)b * a.n_head + h) * L + i]) : 0.f;
                float ms = 1.f;
                if (a.drop.p > 0.f) ms = drop_scale(a.drop, ((unsigned long long)(b * a.n_head + h) * L + i) * L + jc);
                const float dp = dot_row<DH>(vj, gi, dh, v4) * ms;
                axpy_row<DH>(dk, p * (dp - delta) * a.scale, qi, dh, v4);
                axpy_row<DH>(dv, p * ms, gi, dh, v4);
            }
            if (j < L) {
                float* ok = a.dk + ((long)b * L + j) * a.ld_d + hc;
                float* ov = a.dv + ((long)b * L + j) * a.ld_d + hc;
#pragma unroll
                for (int d = 0; d < DH; d += 4) {
                    if (d < dh) {
                        st4g(ok + d, make_float4(dk[d], dk[d + 1], dk[d + 2], dk[d + 3]), dh - d, v4);
                        st4g(ov + d, make_float4(dv[d], dv[d + 1], dv[d + 2], dv[d + 3]), dh - d, v4);
                    }
                }
            }
        }
    }
}

// capacity of the per-thread vectors for a head width
static int cap_of(int d_head) { return d_head <= 8 ? 8 : d_head <= 16 ? 16 : d_head <= 32 ? 32 : d_head <= 64 ? 64 : d_head <= 128 ? 128 : 256; }

}  // namespace

int t4r_reduce_partials_launch(hipStream_t st, const float* part, int nblocks, float* o0, int n0, int a0,
                               float* o1, int n1, int a1, float* o2, int n2, int a2);
extern "C" int t4r_xlnet_attn_bwd_blocks(int B);

// any L >= 1; head widths up to 256 (16-byte requests when the width is a multiple of 4, element by element otherwise; above 64
// the per-thread vectors spill to scratch: slow, but these are the shapes nothing else takes)
int t4r_xlnet_attn_long_ok(int L, int d_head) { return L >= 1 && d_head >= 1 && d_head <= 256; }
int t4r_xlnet_attn_long_fwd(hipStream_t st, const float* q, const float* k, const float* v, const float* kr, const float* rw,
                            const float* rr, float* out, float* lse, int B, int L, int n_head, int d_head, float scale,
                            long kr_bstride, DropCfg drop, const int* key_len) {
    LongArgs a{};
    a.q = q; a.k = k; a.v = v; a.kr = kr; a.rw = rw; a.rr = rr; a.o = out; a.lse_o = lse;
    a.B = B; a.L = L; a.n_head = n_head; a.dh = d_head; a.scale = scale; a.kr_bstride = kr_bstride; a.drop = drop; a.key_len = key_len;
    const dim3 grid(B < 8192 ? B : 8192, n_head), block(64);
    if (!t4r_xlnet_attn_long_ok(L, d_head)) { t4r_set_error("xlnet_attn: d_head must be at most 256"); return -1; }
    switch (cap_of(d_head)) {
        case 8: hipLaunchKernelGGL(attn_long_fwd_kernel<8>, grid, block, 0, st, a); break;
        case 16: hipLaunchKernelGGL(attn_long_fwd_kernel<16>, grid, block, 0, st, a); break;
        case 32: hipLaunchKernelGGL(attn_long_fwd_kernel<32>, grid, block, 0, st, a); break;
        case 64: hipLaunchKernelGGL(attn_long_fwd_kernel<64>, grid, block, 0, st, a); break;
        case 128: hipLaunchKernelGGL(attn_long_fwd_kernel<128>, grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL(attn_long_fwd_kernel<256>, grid, block, 0, st, a); break;
    }
    T4R_LAUNCH_CHECK();
    return 0;
}

// part: blocks x (2 L D + 2 D) partial rows (the head of t4r_xlnet_attn_bwd_ws_floats' buffer); delta: B n_head L floats behind them
int t4r_xlnet_attn_long_bwd(hipStream_t st, const float* q, const float* k, const float* v, const float* kr, const float* rw,
                            const float* rr, const float* out, const float* lse, const float* dout, float* dq, float* dk,
                            float* dv, float* part, float* delta, float* dkr, float* d_rw, float* d_rr, int B, int L, int n_head,
                            int d_head, float scale, long kr_bstride, DropCfg drop, const int* key_len) {
    const int D = n_head * d_head, nblocks = t4r_xlnet_attn_bwd_blocks(B);
    LongArgs a{};
    a.q = q; a.k = k; a.v = v; a.kr = kr; a.rw = rw; a.rr = rr; a.out = out; a.lse = lse; a.dout = dout;
    a.dq = dq; a.dk = dk; a.dv = dv; a.part = part; a.delta = delta;
    a.dkr_b = kr_bstride > 0 ? dkr : nullptr;
    a.B = B; a.L = L; a.n_head = n_head; a.dh = d_head; a.scale = scale; a.kr_bstride = kr_bstride; a.drop = drop; a.key_len = key_len;
    const dim3 grid(nblocks, n_head), block(64);
    if (!t4r_xlnet_attn_long_ok(L, d_head)) { t4r_set_error("xlnet_attn_bwd: d_head must be at most 256"); return -1; }
    const bool shared = kr_bstride == 0;
#define T4R_LONG_BWD(DHV)                                                                              \
    hipLaunchKernelGGL(attn_long_bwd_rows_kernel<DHV>, grid, block, 0, st, a);                         \
    hipLaunchKernelGGL(attn_long_bwd_keys_kernel<DHV>, grid, block, 0, st, a);                         \
    if (shared) hipLaunchKernelGGL((attn_long_bwd_rel_kernel<DHV, true>), grid, block, 0, st, a);      \
    else hipLaunchKernelGGL((attn_long_bwd_rel_kernel<DHV, false>), grid, block, 0, st, a);
    switch (cap_of(d_head)) {
        case 8: T4R_LONG_BWD(8) break;
        case 16: T4R_LONG_BWD(16) break;
        case 32: T4R_LONG_BWD(32) break;
        case 64: T4R_LONG_BWD(64) break;
        case 128: T4R_LONG_BWD(128) break;
        default: T4R_LONG_BWD(256) break;
    }
#undef T4R_LONG_BWD
    T4R_LAUNCH_CHECK();
    // d k_r overwritten (shared k_r: summed over the workgroups here), bias gradients accumulated
    return t4r_reduce_partials_launch(st, part, nblocks, shared ? dkr : nullptr, 2 * L * D, 0, d_rw, D, 1, d_rr, D, 1);
}

// ---- scaled-dot-product core (GPT-2 / BERT): any L, d_head a multiple of 4 up to 128
int t4r_mha_long_ok(int L, int d_head) { return t4r_xlnet_attn_long_ok(L, d_head); }
int t4r_mha_long_fwd(hipStream_t st, const float* q, const float* k, const float* v, long ld, float* out, long ld_out, float* lse,
                     int B, int L, int n_head, int d_head, float scale, int causal, DropCfg drop, const int* key_len) {
    MhaLongArgs a{};
    a.q = q; a.k = k; a.v = v; a.o = out; a.lse_o = lse; a.ld = ld; a.ld_out = ld_out;
    a.B = B; a.L = L; a.n_head = n_head; a.dh = d_head; a.causal = causal; a.scale = scale; a.drop = drop; a.key_len = key_len;
    const dim3 grid(B < 8192 ? B : 8192, n_head), block(64);
    switch (cap_of(d_head)) {
        case 8: hipLaunchKernelGGL(mha_long_fwd_kernel<8>, grid, block, 0, st, a); break;
        case 16: hipLaunchKernelGGL(mha_long_fwd_kernel<16>, grid, block, 0, st, a); break;
        case 32: hipLaunchKernelGGL(mha_long_fwd_kernel<32>, grid, block, 0, st, a); break;
        case 64: hipLaunchKernelGGL(mha_long_fwd_kernel<64>, grid, block, 0, st, a); break;
        case 128: hipLaunchKernelGGL(mha_long_fwd_kernel<128>, grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL(mha_long_fwd_kernel<256>, grid, block, 0, st, a); break;
    }
    T4R_LAUNCH_CHECK();
    return 0;
}
int t4r_mha_long_bwd(hipStream_t st, const float* q, const float* k, const float* v, long ld, const float* out, const float* dout,
                     long ld_out, const float* lse, float* dq, float* dk, float* dv, long ld_d, int B, int L, int n_head,
                     int d_head, float scale, int causal, DropCfg drop, const int* key_len) {
    MhaLongArgs a{};
    a.q = q; a.k = k; a.v = v; a.out = out; a.dout = dout; a.lse = lse; a.dq = dq; a.dk = dk; a.dv = dv;
    a.ld = ld; a.ld_out = ld_out; a.ld_d = ld_d;
    a.B = B; a.L = L; a.n_head = n_head; a.dh = d_head; a.causal = causal; a.scale = scale; a.drop = drop; a.key_len = key_len;
    const dim3 grid(B < 8192 ? B : 8192, n_head), block(64);
#define T4R_MHA_LONG_BWD(DHV)                                                        \
    hipLaunchKernelGGL(mha_long_bwd_rows_kernel<DHV>, grid, block, 0, st, a);        \
    hipLaunchKernelGGL(mha_long_bwd_keys_kernel<DHV>, grid, block, 0, st, a);
    switch (cap_of(d_head)) {
        case 8: T4R_MHA_LONG_BWD(8) break;
        case 16: T4R_MHA_LONG_BWD(16) break;
        case 32: T4R_MHA_LONG_BWD(32) break;
        case 64: T4R_MHA_LONG_BWD(64) break;
        case 128: T4R_MHA_LONG_BWD(128) break;
        default: T4R_MHA_LONG_BWD(256) break;
    }
#undef T4R_MHA_LONG_BWD
    T4R_LAUNCH_CHECK();
    return 0;
}
