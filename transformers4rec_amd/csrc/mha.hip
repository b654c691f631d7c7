// Scaled-dot-product multi-head attention core (forward + backward) for the GPT-2 (causal) and
// BERT (bidirectional) transformer blocks of the session-sequence path.
//
// Restates the attention arithmetic of the third-party HF models the reference instantiates
// through transformers4rec/config/transformer.py:218-260 (GPT2Config.build) and :493-534
// (BertConfig.build) and calls at transformers4rec/torch/block/transformer.py:179-199:
//   HF gpt2/modeling_gpt2.py eager_attention_forward :54-72   softmax(q k^T / sqrt(dh) + causal) v
//   HF bert/modeling_bert.py BertSelfAttention (same form, no mask: the reference passes none)
// As for XLNet, NO padding mask is applied (reference semantics); GPT-2's
// causal mask is internal to HF (the reference's tril head_mask is filtered out by HF 5.x and
// was a numerical no-op before).
//
// MI355X design: sequences are short (L = 50..100, d_head 32..64): one workgroup per
// (session, head); lane = query row (forward, backward phase 1) or key row (backward phase 2);
// the head's K, V (and Q, dO in the backward) slices live in LDS with a +4 float row pad, rows
// read as wave-uniform broadcasts.  Forward = one online-softmax pass saving the row
// log-sum-exp; backward recomputes the probabilities twice (once per phase) instead of
// storing an [L, L] tile, so LDS holds only 4 * L * (dh + 4) floats (109 KB at L=100, dh=64).
#include "t4r_common.h"

#define MHA_PAD 4

// mha_mfma.hip: matrix-core kernels for d_head 32 / 64
bool t4r_mha_mfma_ok(int L, int d_head, long ld, long ld_out, long ld_d);
int t4r_mha_mfma_fwd(hipStream_t st, const float* q, const float* k, const float* v, long ld, float* out,
                     long ld_out, float* lse, int B, int L, int n_head, int d_head, float scale, int causal,
                     DropCfg drop, const int* key_len);
int t4r_mha_mfma_bwd(hipStream_t st, const float* q, const float* k, const float* v, long ld, const float* out,
                     const float* dout, long ld_out, const float* lse, float* dq, float* dk, float* dv, long ld_d,
                     int B, int L, int n_head, int d_head, float scale, int causal, DropCfg drop, const int* key_len);

template <int DH>
__global__ __launch_bounds__(128) void mha_fwd_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, long ld,
    float* __restrict__ out, long ld_out, float* __restrict__ lse, int B, int L, int n_head, float scale,
    int causal, DropCfg drop, const int* key_len) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDH = DH + MHA_PAD;
    float* Ks = smem;             // [L][LDH]
    float* Vs = Ks + L * LDH;     // [L][LDH]
    const int b = blockIdx.x, h = blockIdx.y;
    const int hc = h * DH;
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int i = tid; i < L * (DH / 4); i += nthr) {
        const int r = i / (DH / 4), c = (i % (DH / 4)) * 4;
        const long g = ((long)b * L + r) * ld + hc + c;
        *reinterpret_cast<float4*>(Ks + r * LDH + c) = *reinterpret_cast<const float4*>(k + g);
        *reinterpret_cast<float4*>(Vs + r * LDH + c) = *reinterpret_cast<const float4*>(v + g);
    }
    __syncthreads();
    const int i = tid;
    if (i >= L) return;
    float qi[DH], o[DH];
    const float* qrow = q + ((long)b * L + i) * ld + hc;
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
        const float4 t = *reinterpret_cast<const float4*>(qrow + d);
        qi[d] = t.x; qi[d + 1] = t.y; qi[d + 2] = t.z; qi[d + 3] = t.w;
        o[d] = o[d + 1] = o[d + 2] = o[d + 3] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    // opt-in padding mask: keys >= key_len[b] are masked for every query (HF adds finfo.min to their scores)
    const int klen = key_len ? max(1, min(L, key_len[b])) : L;
    const int jend = min(causal ? i + 1 : L, klen);
    for (int j = 0; j < jend; ++j) {
        const float* kj = Ks + j * LDH;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            const float4 a = *reinterpret_cast<const float4*>(kj + d);
            s += qi[d] * a.x + qi[d + 1] * a.y + qi[d + 2] * a.z + qi[d + 3] * a.w;
        }
        s *= scale;
        const float mn = fmaxf(m, s);
        const float alpha = __expf(m - mn);
        const float pj = __expf(s - mn);
        l = l * alpha + pj;
        float pd = pj;
        if (drop.p > 0.f) pd *= drop_scale(drop, ((unsigned long long)(b * n_head + h) * L + i) * L + j);
        const float* vj = Vs + j * LDH;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            const float4 a = *reinterpret_cast<const float4*>(vj + d);
            o[d] = o[d] * alpha + pd * a.x;
            o[d + 1] = o[d + 1] * alpha + pd * a.y;
            o[d + 2] = o[d + 2] * alpha + pd * a.z;
            o[d + 3] = o[d + 3] * alpha + pd * a.w;
        }
        m = mn;
    }
    const float inv = 1.f / l;
    float* orow = out + ((long)b * L + i) * ld_out + hc;
#pragma unroll
    for (int d = 0; d < DH; d += 4)
        *reinterpret_cast<float4*>(orow + d) = make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
    lse[((long)b * n_head + h) * L + i] = m + __logf(l);
}

template <int DH>
__global__ __launch_bounds__(128) void mha_bwd_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, long ld,
    const float* __restrict__ out, const float* __restrict__ dout, long ld_out,
    const float* __restrict__ lse, float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
    long ld_d, int B, int L, int n_head, float scale, int causal, DropCfg drop, const int* key_len) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDH = DH + MHA_PAD;
    float* Ks = smem;              // [L][LDH]
    float* Vs = Ks + L * LDH;
    float* Qs = Vs + L * LDH;
    float* dOs = Qs + L * LDH;
    float* Ds = dOs + L * LDH;     // [L]  D_i = dO_i . O_i
    float* Ls = Ds + L;            // [L]  lse_i
    const int b = blockIdx.x, h = blockIdx.y;
    const int hc = h * DH;
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int i = tid; i < L * (DH / 4); i += nthr) {
        const int r = i / (DH / 4), c = (i % (DH / 4)) * 4;
        const long g = ((long)b * L + r) * ld + hc + c;
        *reinterpret_cast<float4*>(Ks + r * LDH + c) = *reinterpret_cast<const float4*>(k + g);
        *reinterpret_cast<float4*>(Vs + r * LDH + c) = *reinterpret_cast<const float4*>(v + g);
        *reinterpret_cast<float4*>(Qs + r * LDH + c) = *reinterpret_cast<const float4*>(q + g);
        *reinterpret_cast<float4*>(dOs + r * LDH + c) =
            *reinterpret_cast<const float4*>(dout + ((long)b * L + r) * ld_out + hc + c);
    }
    __syncthreads();
    const unsigned long long mbase = (unsigned long long)(b * n_head + h) * L;
    // ---- phase 1: lane = query row i  ->  D_i, d q_i
    if (tid < L) {
        const int i = tid;
        const float* orow = out + ((long)b * L + i) * ld_out + hc;
        const float* qi = Qs + i * LDH;
        const float* gi = dOs + i * LDH;
        float Di = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) Di += gi[d] * orow[d];
        const float lse_i = lse[((long)b * n_head + h) * L + i];
        Ds[i] = Di;
        Ls[i] = lse_i;
        float g[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) g[d] = 0.f;
        const int jend = min(causal ? i + 1 : L, key_len ? max(1, min(L, key_len[b])) : L);
        for (int j = 0; j < jend; ++j) {
            const float* kj = Ks + j * LDH;
            const float* vj = Vs + j * LDH;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < DH; d += 4) {
                const float4 a = *reinterpret_cast<const float4*>(kj + d);
                const float4 e = *reinterpret_cast<const float4*>(vj + d);
                const float4 qq = *reinterpret_cast<const float4*>(qi + d);
                const float4 gg = *reinterpret_cast<const float4*>(gi + d);
                s += qq.x * a.x + qq.y * a.y + qq.z * a.z + qq.w * a.w;
                dp += gg.x * e.x + gg.y * e.y + gg.z * e.z + gg.w * e.w;
            }
            const float p = __expf(s * scale - lse_i);
            const float msk = drop.p > 0.f ? drop_scale(drop, (mbase + i) * L + j) : 1.f;
            const float ds = p * (dp * msk - Di) * scale;
#pragma unroll
            for (int d = 0; d < DH; d += 4) {
                const float4 a = *reinterpret_cast<const float4*>(kj + d);
                g[d] += ds * a.x; g[d + 1] += ds * a.y; g[d + 2] += ds * a.z; g[d + 3] += ds * a.w;
            }
        }
        float* dqrow = dq + ((long)b * L + i) * ld_d + hc;
#pragma unroll
        for (int d = 0; d < DH; d += 4)
            *reinterpret_cast<float4*>(dqrow + d) = make_float4(g[d], g[d + 1], g[d + 2], g[d + 3]);
    }
    __syncthreads();
    // ---- phase 2: lane = key row j  ->  d k_j = sum_i dS_ij q_i ; d v_j = sum_i Pdrop_ij dO_i
    if (tid < L) {
        const int j = tid;
        const float* kj = Ks + j * LDH;
        const float* vj = Vs + j * LDH;
        float gk[DH], gv[DH];
#pragma unroll
        for (int d = 0; d < DH; ++d) gk[d] = gv[d] = 0.f;
        for (int i = causal ? j : 0; i < L; ++i) {
            const float* qi = Qs + i * LDH;
            const float* gi = dOs + i * LDH;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < DH; d += 4) {
                const float4 a = *reinterpret_cast<const float4*>(kj + d);
                const float4 e = *reinterpret_cast<const float4*>(vj + d);
                const float4 qq = *reinterpret_cast<const float4*>(qi + d);
                const float4 gg = *reinterpret_cast<const float4*>(gi + d);
                s += qq.x * a.x + qq.y * a.y + qq.z * a.z + qq.w * a.w;
                dp += gg.x * e.x + gg.y * e.y + gg.z * e.z + gg.w * e.w;
            }
            const float p = (key_len && j >= max(1, min(L, key_len[b]))) ? 0.f : __expf(s * scale - Ls[i]);
            const float msk = drop.p > 0.f ? drop_scale(drop, (mbase + i) * L + j) : 1.f;
            const float ds = p * (dp * msk - Ds[i]) * scale;
            const float pd = p * msk;
#pragma unroll
            for (int d = 0; d < DH; d += 4) {
                const float4 qq = *reinterpret_cast<const float4*>(qi + d);
                const float4 gg = *reinterpret_cast<const float4*>(gi + d);
                gk[d] += ds * qq.x; gk[d + 1] += ds * qq.y; gk[d + 2] += ds * qq.z; gk[d + 3] += ds * qq.w;
                gv[d] += pd * gg.x; gv[d + 1] += pd * gg.y; gv[d + 2] += pd * gg.z; gv[d + 3] += pd * gg.w;
            }
        }
        float* dkrow = dk + ((long)b * L + j) * ld_d + hc;
        float* dvrow = dv + ((long)b * L + j) * ld_d + hc;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            *reinterpret_cast<float4*>(dkrow + d) = make_float4(gk[d], gk[d + 1], gk[d + 2], gk[d + 3]);
            *reinterpret_cast<float4*>(dvrow + d) = make_float4(gv[d], gv[d + 1], gv[d + 2], gv[d + 3]);
        }
    }
}

template <int DH>
static int mha_fwd_launch(hipStream_t st, const float* q, const float* k, const float* v, long ld, float* out,
                          long ld_out, float* lse, int B, int L, int n, float scale, int causal, DropCfg dc, const int* key_len) {
    const size_t smem = (size_t)2 * L * (DH + MHA_PAD) * sizeof(float);
    static T4rLdsAttr attr;
    t4r_ensure_dynamic_lds((const void*)mha_fwd_kernel<DH>, smem, attr);
    hipLaunchKernelGGL(mha_fwd_kernel<DH>, dim3(B, n), dim3(L <= 64 ? 64 : 128), smem, st, q, k, v, ld, out, ld_out,
                       lse, B, L, n, scale, causal, dc, key_len);
    T4R_LAUNCH_CHECK();
    return 0;
}
template <int DH>
static int mha_bwd_launch(hipStream_t st, const float* q, const float* k, const float* v, long ld, const float* out,
                          const float* dout, long ld_out, const float* lse, float* dq, float* dk, float* dv,
                          long ld_d, int B, int L, int n, float scale, int causal, DropCfg dc, const int* key_len) {
    const size_t smem = ((size_t)4 * L * (DH + MHA_PAD) + 2 * L) * sizeof(float);
    static T4rLdsAttr attr;
    t4r_ensure_dynamic_lds((const void*)mha_bwd_kernel<DH>, smem, attr);
    hipLaunchKernelGGL(mha_bwd_kernel<DH>, dim3(B, n), dim3(L <= 64 ? 64 : 128), smem, st, q, k, v, ld, out, dout,
                       ld_out, lse, dq, dk, dv, ld_d, B, L, n, scale, causal, dc, key_len);
    T4R_LAUNCH_CHECK();
    return 0;
}

// q,k,v: rows of `ld` floats (ld = 3*D for GPT-2's fused c_attn output, D for separate buffers),
// head h at columns [h*d_head, (h+1)*d_head).  out/dout rows of ld_out floats.  lse [B, n, L].
// general kernels (xlnet_attn_long.hip): any L, d_head up to 256
int t4r_mha_long_ok(int L, int d_head);
int t4r_mha_long_fwd(hipStream_t st, const float* q, const float* k, const float* v, long ld, float* out, long ld_out, float* lse,
                     int B, int L, int n_head, int d_head, float scale, int causal, DropCfg drop, const int* key_len);
int t4r_mha_long_bwd(hipStream_t st, const float* q, const float* k, const float* v, long ld, const float* out, const float* dout,
                     long ld_out, const float* lse, float* dq, float* dk, float* dv, long ld_d, int B, int L, int n_head,
                     int d_head, float scale, int causal, DropCfg drop, const int* key_len);
static bool mha_short_ok(int L, int d_head) { return L <= 128 && (d_head == 16 || d_head == 32 || d_head == 64); }

extern "C" int t4r_mha_fwd(void* stream, const float* q, const float* k, const float* v, long ld, float* out,
                           long ld_out, float* lse, int B, int L, int n_head, int d_head, int causal,
                           float drop_p, unsigned long long seed, unsigned long long ctr_hi, const int* key_len) {
    if (B == 0) return 0;
    T4R_CHECK_ARG(L >= 1, "mha: L must be at least 1");
    T4R_CHECK_ARG(ld % 4 == 0 && ld_out % 4 == 0, "mha: row strides must be multiples of 4");
    const float scale = 1.0f / sqrtf((float)d_head);
    const DropCfg dc = make_drop(drop_p, seed, ctr_hi);
    hipStream_t st = (hipStream_t)stream;
    if (!mha_short_ok(L, d_head)) {      // beyond 128 positions, or a head width the LDS kernels have no instance for
        T4R_CHECK_ARG(t4r_mha_long_ok(L, d_head), "mha: d_head must be at most 256");
        return t4r_mha_long_fwd(st, q, k, v, ld, out, ld_out, lse, B, L, n_head, d_head, scale, causal, dc, key_len);
    }
    if (t4r_mha_mfma_ok(L, d_head, ld, ld_out, 0))
        return t4r_mha_mfma_fwd(st, q, k, v, ld, out, ld_out, lse, B, L, n_head, d_head, scale, causal, dc, key_len);
    switch (d_head) {
        case 16: return mha_fwd_launch<16>(st, q, k, v, ld, out, ld_out, lse, B, L, n_head, scale, causal, dc, key_len);
        case 32: return mha_fwd_launch<32>(st, q, k, v, ld, out, ld_out, lse, B, L, n_head, scale, causal, dc, key_len);
        case 64: return mha_fwd_launch<64>(st, q, k, v, ld, out, ld_out, lse, B, L, n_head, scale, causal, dc, key_len);
    }
    t4r_set_error("mha: d_head must be 16, 32 or 64");
    return -1;
}

extern "C" int t4r_mha_bwd(void* stream, const float* q, const float* k, const float* v, long ld,
                           const float* out, const float* dout, long ld_out, const float* lse, float* dq,
                           float* dk, float* dv, long ld_d, int B, int L, int n_head, int d_head, int causal,
                           float drop_p, unsigned long long seed, unsigned long long ctr_hi, const int* key_len) {
    if (B == 0) return 0;
    T4R_CHECK_ARG(L >= 1, "mha: L must be at least 1");
    T4R_CHECK_ARG(ld % 4 == 0 && ld_out % 4 == 0 && ld_d % 4 == 0, "mha: row strides must be multiples of 4");
    const float scale = 1.0f / sqrtf((float)d_head);
    const DropCfg dc = make_drop(drop_p, seed, ctr_hi);
    hipStream_t st = (hipStream_t)stream;
    if (!mha_short_ok(L, d_head)) {
        T4R_CHECK_ARG(t4r_mha_long_ok(L, d_head), "mha: d_head must be at most 256");
        return t4r_mha_long_bwd(st, q, k, v, ld, out, dout, ld_out, lse, dq, dk, dv, ld_d, B, L, n_head, d_head, scale, causal, dc,
                                key_len);
    }
    if (t4r_mha_mfma_ok(L, d_head, ld, ld_out, ld_d))
        return t4r_mha_mfma_bwd(st, q, k, v, ld, out, dout, ld_out, lse, dq, dk, dv, ld_d, B, L, n_head, d_head, scale,
                                causal, dc, key_len);
    switch (d_head) {
        case 16: return mha_bwd_launch<16>(st, q, k, v, ld, out, dout, ld_out, lse, dq, dk, dv, ld_d, B, L, n_head, scale, causal, dc, key_len);
        case 32: return mha_bwd_launch<32>(st, q, k, v, ld, out, dout, ld_out, lse, dq, dk, dv, ld_d, B, L, n_head, scale, causal, dc, key_len);
        case 64: return mha_bwd_launch<64>(st, q, k, v, ld, out, dout, ld_out, lse, dq, dk, dv, ld_d, B, L, n_head, scale, causal, dc, key_len);
    }
    t4r_set_error("mha: d_head must be 16, 32 or 64");
    return -1;
}
