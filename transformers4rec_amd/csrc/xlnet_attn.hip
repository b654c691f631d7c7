// XLNet relative-position bidirectional attention core, forward and backward (gfx950).
//
// Restates HF transformers modeling_xlnet.py (third-party dependency of the reference,
// reached through transformers4rec/torch/block/transformer.py:179-199):
//   rel_attn_core   :95-140    ac = (q + r_w_bias).k ; bd = rel_shift((q + r_r_bias).k_r)
//                              prob = softmax((ac + bd) * 1/sqrt(d_head)) ; out = prob.v
//   rel_shift_bnij  :81-93     for klen == qlen == L:  bd[i, j] = raw[i, j + L - i]
// with the reference's configuration (config/transformer.py:432-482): attn_type "bi", no
// attention mask (padding positions attend and are attended -- SURVEY fact 3), no segment
// term, dropout handled outside.
//
// MI355X design: the sequences are tiny (L ~ 20, d_head 16-32) so this is not MFMA work: one
// workgroup per session, one wave per head, lane i owns query row i.  K, V and the
// batch-independent positional keys k_r [2L, D] (computed ONCE per layer by a GEMM, not per
// batch row as the reference does) sit in LDS with a +4 float row pad (conflict-free 16-byte
// reads for the lane-dependent k_r row j+L-i; k_j / v_j rows are wave-uniform broadcasts).
// Forward is a single online-softmax pass and saves only the row log-sum-exp; backward
// recomputes the probabilities (no [B,n,L,L] tensor ever goes to HBM).
#include "t4r_common.h"
// additive score of a masked key (opt-in padding mask): HF modeling_xlnet.py subtracts 1e30 * attn_mask in fp32
#define T4R_KEY_MASKED (-1e30f)
#include <stdlib.h>

#define ATT_PAD 4
// heads (= waves) per workgroup: 4 for d_head >= 32 so the compiler may use up to 512 VGPRs
// (one head per wave, lane = row: q+r_w, q+r_r, two gradient accumulators of d_head floats each)
template <int DH> struct HeadsPerBlock { static constexpr int v = DH >= 32 ? 4 : 8; };

template <int DH>
__global__ __launch_bounds__(64 * HeadsPerBlock<DH>::v) void xlnet_attn_fwd_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ kr,      // [2L, D]
    const float* __restrict__ r_w_bias, const float* __restrict__ r_r_bias,  // [D]
    float* __restrict__ out,           // [B*L, D]
    float* __restrict__ lse,           // [B, n, L]
    int B, int L, int n_head, float scale, long kr_bstride, DropCfg drop, const int* key_len) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int D = n_head * DH;
    const int LD = D + ATT_PAD;
    float* Ks = smem;                 // [L][LD]
    float* Vs = Ks + L * LD;          // [L][LD]
    float* KRs = Vs + L * LD;         // [2L][LD]
    const int b = blockIdx.x;
    kr += (long)b * kr_bstride;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int dq = D / 4;
    for (int i = tid; i < L * dq; i += nthr) {
        const int r = i / dq, c = (i % dq) * 4;
        *reinterpret_cast<float4*>(Ks + r * LD + c) =
            *reinterpret_cast<const float4*>(k + ((long)b * L + r) * D + c);
        *reinterpret_cast<float4*>(Vs + r * LD + c) =
            *reinterpret_cast<const float4*>(v + ((long)b * L + r) * D + c);
    }
    for (int i = tid; i < 2 * L * dq; i += nthr) {
        const int r = i / dq, c = (i % dq) * 4;
        *reinterpret_cast<float4*>(KRs + r * LD + c) = *reinterpret_cast<const float4*>(kr + (long)r * D + c);
    }
    __syncthreads();
    const int lane = tid & 63;
    {
        const int h = blockIdx.y * HeadsPerBlock<DH>::v + (tid >> 6);   // one head per wave
        const int i = lane;
        if (i >= L || h >= n_head) return;
        const int hc = h * DH;
        float qw[DH], qr[DH], o[DH];
        const float* qrow = q + ((long)b * L + i) * D + hc;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            const float4 t = *reinterpret_cast<const float4*>(qrow + d);
            const float4 bw = *reinterpret_cast<const float4*>(r_w_bias + hc + d);
            const float4 br = *reinterpret_cast<const float4*>(r_r_bias + hc + d);
            qw[d] = t.x + bw.x; qw[d + 1] = t.y + bw.y; qw[d + 2] = t.z + bw.z; qw[d + 3] = t.w + bw.w;
            qr[d] = t.x + br.x; qr[d + 1] = t.y + br.y; qr[d + 2] = t.z + br.z; qr[d + 3] = t.w + br.w;
        }
#pragma unroll
        for (int d = 0; d < DH; ++d) o[d] = 0.f;
        float m = -INFINITY, l = 0.f;
        const int klen = key_len ? key_len[b] : L;
        for (int j = 0; j < L; ++j) {
            const float* kj = Ks + j * LD + hc;
            const float* krp = KRs + (j + L - i) * LD + hc;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < DH; d += 4) {
                const float4 a = *reinterpret_cast<const float4*>(kj + d);
                const float4 c = *reinterpret_cast<const float4*>(krp + d);
                s += qw[d] * a.x + qw[d + 1] * a.y + qw[d + 2] * a.z + qw[d + 3] * a.w;
                s += qr[d] * c.x + qr[d + 1] * c.y + qr[d + 2] * c.z + qr[d + 3] * c.w;
            }
            s *= scale;
            if (j >= klen && j != i) s = T4R_KEY_MASKED;     // opt-in padding mask (HF: score - 1e30 * mask, diagonal kept)
            const float mn = fmaxf(m, s);
            const float alpha = __expf(m - mn);
            const float pj = __expf(s - mn);
            l = l * alpha + pj;
            // attention-probability dropout (HF rel_attn_core :132): the normaliser keeps all terms
            float pd = pj;
            if (drop.p > 0.f) pd *= drop_scale(drop, ((unsigned long long)(b * n_head + h) * L + i) * L + j);
            const float* vj = Vs + j * LD + hc;
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
        float* orow = out + ((long)b * L + i) * D + hc;
#pragma unroll
        for (int d = 0; d < DH; d += 4)
            *reinterpret_cast<float4*>(orow + d) =
                make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
        lse[((long)b * n_head + h) * L + i] = m + __logf(l);
    }
}

// Backward.  Grid-stride over sessions; per block partial sums of the batch-reduced
// gradients (d k_r, d r_w_bias, d r_r_bias) go to a workspace and are summed by
// xlnet_attn_bwd_reduce_kernel (deterministic, no atomics).
template <int DH>
__global__ __launch_bounds__(64 * HeadsPerBlock<DH>::v) void xlnet_attn_bwd_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ kr, const float* __restrict__ r_w_bias,
    const float* __restrict__ r_r_bias, const float* __restrict__ out,
    const float* __restrict__ lse, const float* __restrict__ dout,
    float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
    float* __restrict__ part,          // [grid][2L*D + 2*D]
    float* __restrict__ dkr_b,         // per-batch d k_r [B][2L][D] (kr_bstride > 0) or null
    int B, int L, int n_head, float scale, long kr_bstride, DropCfg drop, const int* key_len) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int D = n_head * DH;
    const int LD = D + ATT_PAD;
    const int LS = L + 1;
    float* Ks = smem;                        // [L][LD]
    float* Vs = Ks + L * LD;                 // [L][LD]
    float* Qs = Vs + L * LD;                 // [L][LD]
    float* dOs = Qs + L * LD;                // [L][LD]
    float* KRs = dOs + L * LD;               // [2L][LD]
    float* dKR = KRs + 2 * L * LD;           // [2L][LD]   block accumulator (shared k_r only)
    float* dSs = dKR + (kr_bstride > 0 ? 0 : 2 * L * LD);   // [n][L][LS]
    float* Ps = dSs + n_head * L * LS;       // [n][L][LS]
    float* dRW = Ps + n_head * L * LS;       // [D]  block accumulators of d r_w_bias / d r_r_bias
    float* dRR = dRW + D;                    // [D]
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63;
    const int h = blockIdx.y * HeadsPerBlock<DH>::v + (tid >> 6);       // one head per wave
    const bool hvalid = h < n_head;
    const int dq4 = D / 4;
    if (kr_bstride == 0) {
        for (int i = tid; i < 2 * L * dq4; i += nthr) {
            const int r = i / dq4, c = (i % dq4) * 4;
            *reinterpret_cast<float4*>(KRs + r * LD + c) = *reinterpret_cast<const float4*>(kr + (long)r * D + c);
            *reinterpret_cast<float4*>(dKR + r * LD + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    for (int i = tid; i < 2 * D; i += nthr) dRW[i] = 0.f;

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();  // previous session's phases done before the tiles are overwritten
        if (kr_bstride > 0) {   // per-session positional keys (pos_emb dropout): reload
            for (int i = tid; i < 2 * L * dq4; i += nthr) {
                const int r = i / dq4, c = (i % dq4) * 4;
                *reinterpret_cast<float4*>(KRs + r * LD + c) =
                    *reinterpret_cast<const float4*>(kr + (long)b * kr_bstride + (long)r * D + c);
            }
        }
        for (int i = tid; i < L * dq4; i += nthr) {
            const int r = i / dq4, c = (i % dq4) * 4;
            const long g = ((long)b * L + r) * D + c;
            *reinterpret_cast<float4*>(Ks + r * LD + c) = *reinterpret_cast<const float4*>(k + g);
            *reinterpret_cast<float4*>(Vs + r * LD + c) = *reinterpret_cast<const float4*>(v + g);
            *reinterpret_cast<float4*>(Qs + r * LD + c) = *reinterpret_cast<const float4*>(q + g);
            *reinterpret_cast<float4*>(dOs + r * LD + c) = *reinterpret_cast<const float4*>(dout + g);
        }
        __syncthreads();
        // ---- phase 1: lane = query row i
        if (hvalid) {
            const int i = lane;
            const int hc = h * DH;
            float ga[DH], gb[DH];
#pragma unroll
            for (int d = 0; d < DH; ++d) ga[d] = gb[d] = 0.f;
            if (lane < L && drop.p > 0.f) {
                // keep-scales first (low register pressure here), parked in this row's P slot
                float* prow0 = Ps + (h * L + lane) * LS;
#pragma unroll 1
                for (int j = 0; j < L; ++j)
                    prow0[j] = drop_scale(drop, ((unsigned long long)(b * n_head + h) * L + lane) * L + j);
            }
            if (lane < L) {
            float qw[DH], qr[DH];
            const float* go = dOs + i * LD + hc;   // d out row stays in LDS (register budget)
            const float* orow = out + ((long)b * L + i) * D + hc;
            float Di = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) {
                const float qv = Qs[i * LD + hc + d];
                qw[d] = qv + r_w_bias[hc + d];
                qr[d] = qv + r_r_bias[hc + d];
                Di += go[d] * orow[d];
            }
            const float lse_i = lse[((long)b * n_head + h) * L + i];
            float* dsrow = dSs + (h * L + i) * LS;
            float* prow = Ps + (h * L + i) * LS;
#pragma unroll 1
            for (int j = 0; j < L; ++j) {
                const float* kj = Ks + j * LD + hc;
                const float* krp = KRs + (j + L - i) * LD + hc;
                const float* vj = Vs + j * LD + hc;
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < DH; d += 4) {
                    const float4 a = *reinterpret_cast<const float4*>(kj + d);
                    const float4 c = *reinterpret_cast<const float4*>(krp + d);
                    const float4 e = *reinterpret_cast<const float4*>(vj + d);
                    const float4 gq = *reinterpret_cast<const float4*>(go + d);
                    s += qw[d] * a.x + qw[d + 1] * a.y + qw[d + 2] * a.z + qw[d + 3] * a.w;
                    s += qr[d] * c.x + qr[d + 1] * c.y + qr[d + 2] * c.z + qr[d + 3] * c.w;
                    dp += gq.x * e.x + gq.y * e.y + gq.z * e.z + gq.w * e.w;
                }
                const float p = (key_len && j >= key_len[b] && j != i) ? 0.f : __expf(s * scale - lse_i);
                const float msk = drop.p > 0.f ? prow[j] : 1.f;
                const float ds = p * (dp * msk - Di) * scale;
                prow[j] = p * msk;    // dropped probability: what multiplied v in the forward
                dsrow[j] = ds;
#pragma unroll
                for (int d = 0; d < DH; d += 4) {
                    const float4 a = *reinterpret_cast<const float4*>(kj + d);
                    const float4 c = *reinterpret_cast<const float4*>(krp + d);
                    ga[d] += ds * a.x; ga[d + 1] += ds * a.y; ga[d + 2] += ds * a.z; ga[d + 3] += ds * a.w;
                    gb[d] += ds * c.x; gb[d + 1] += ds * c.y; gb[d + 2] += ds * c.z; gb[d + 3] += ds * c.w;
                }
            }
            float* dqrow = dq + ((long)b * L + i) * D + hc;
#pragma unroll
            for (int d = 0; d < DH; d += 4)
                *reinterpret_cast<float4*>(dqrow + d) = make_float4(
                    ga[d] + gb[d], ga[d + 1] + gb[d + 1], ga[d + 2] + gb[d + 2], ga[d + 3] + gb[d + 3]);
            }
            // d r_w_bias[h] += sum_i ga_i ; d r_r_bias[h] += sum_i gb_i  (this wave owns head h)
#pragma unroll
            for (int d = 0; d < DH; ++d) {
                const float sw = wave_sum(ga[d]);
                const float sr = wave_sum(gb[d]);
                if (lane == 0) { dRW[hc + d] += sw; dRR[hc + d] += sr; }
            }
        }
        __syncthreads();
        // ---- phase 2: lane = key row j :  dk_j = sum_i dS_ij (q_i + r_w) ; dv_j = sum_i P_ij dO_i
        if (hvalid && lane < L) {
            const int j = lane;
            const int hc = h * DH;
            float gk[DH], gv[DH];
#pragma unroll
            for (int d = 0; d < DH; ++d) gk[d] = gv[d] = 0.f;
            for (int i = 0; i < L; ++i) {
                const float ds = dSs[(h * L + i) * LS + j];
                const float p = Ps[(h * L + i) * LS + j];
                const float* qi = Qs + i * LD + hc;
                const float* gi = dOs + i * LD + hc;
#pragma unroll
                for (int d = 0; d < DH; d += 4) {
                    const float4 a = *reinterpret_cast<const float4*>(qi + d);
                    const float4 bw = *reinterpret_cast<const float4*>(r_w_bias + hc + d);
                    const float4 e = *reinterpret_cast<const float4*>(gi + d);
                    gk[d] += ds * (a.x + bw.x); gk[d + 1] += ds * (a.y + bw.y);
                    gk[d + 2] += ds * (a.z + bw.z); gk[d + 3] += ds * (a.w + bw.w);
                    gv[d] += p * e.x; gv[d + 1] += p * e.y; gv[d + 2] += p * e.z; gv[d + 3] += p * e.w;
                }
            }
            float* dkrow = dk + ((long)b * L + j) * D + hc;
            float* dvrow = dv + ((long)b * L + j) * D + hc;
#pragma unroll
            for (int d = 0; d < DH; d += 4) {
                *reinterpret_cast<float4*>(dkrow + d) = make_float4(gk[d], gk[d + 1], gk[d + 2], gk[d + 3]);
                *reinterpret_cast<float4*>(dvrow + d) = make_float4(gv[d], gv[d + 1], gv[d + 2], gv[d + 3]);
            }
        }
        // ---- phase 3: d k_r[p] += sum_i dS[i, p - L + i] (q_i + r_r)   (p = j + L - i)
        if (hvalid) {
            const int hc = h * DH;
            for (int p = lane; p < 2 * L; p += 64) {
                float g[DH];
#pragma unroll
                for (int d = 0; d < DH; ++d) g[d] = 0.f;
                for (int i = 0; i < L; ++i) {
                    const int j = p - L + i;
                    if (j < 0 || j >= L) continue;
                    const float ds = dSs[(h * L + i) * LS + j];
                    const float* qi = Qs + i * LD + hc;
#pragma unroll
                    for (int d = 0; d < DH; ++d) g[d] += ds * (qi[d] + r_r_bias[hc + d]);
                }
                if (dkr_b) {
                    float* dst = dkr_b + ((long)b * 2 * L + p) * D + hc;
#pragma unroll
                    for (int d = 0; d < DH; ++d) dst[d] = g[d];
                } else {
                    float* acc = dKR + p * LD + hc;
#pragma unroll
                    for (int d = 0; d < DH; ++d) acc[d] += g[d];
                }
            }
        }
    }
    __syncthreads();
    // block partials -> workspace
    float* mypart = part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (2 * L * D + 2 * D);
    if (kr_bstride == 0)
        for (int i = tid; i < 2 * L * D; i += nthr) mypart[i] = dKR[(i / D) * LD + (i % D)];
    for (int i = tid; i < 2 * D; i += nthr) mypart[2 * L * D + i] = dRW[i];
}

// ------------------------------------------------------------------------------------------------
// Backward, pair-decomposed (used when L <= 32): ONE WAVE per (session, head), all 64 lanes busy.
//   stage A: lanes stride over the L*L (i, j) pairs: recompute p_ij, form dS_ij and the dropped
//            probability, park both in LDS;
//   stage B: lanes stride over (row, 4-column chunk) output items and contract dS / P against the
//            LDS-resident rows:  dq_i = sum_j dS_ij (k_j + k_r[j+L-i]),  dk_j = sum_i dS_ij (q_i + r_w),
//            dv_j = sum_i Pd_ij dO_i,  dk_r[p] = sum_i dS_{i,p-L+i} (q_i + r_r).
// The lane-per-row kernel above keeps 20 of 64 lanes busy at 4 waves/CU (230 us per layer at C2);
// this form needs ~21 KB LDS per wave (7 waves/CU) and every lane works.
#define PAIRS_NKR 8
template <int DH>
__global__ __launch_bounds__(64) void xlnet_attn_bwd_pairs_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ kr, const float* __restrict__ r_w_bias, const float* __restrict__ r_r_bias,
    const float* __restrict__ out, const float* __restrict__ lse, const float* __restrict__ dout,
    float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv, float* __restrict__ part,
    float* __restrict__ dkr_b, int B, int L, int n_head, float scale, long kr_bstride, DropCfg drop, const int* key_len) {
    constexpr int LDH = DH + 4;
    constexpr int C = DH / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LS = L + 1;
    float* Qw = smem;                  // [L][LDH]   q + r_w_bias
    float* Ks = Qw + L * LDH;          // [L][LDH]
    float* Vs = Ks + L * LDH;          // [L][LDH]
    float* dOs = Vs + L * LDH;         // [L][LDH]
    float* KRs = dOs + L * LDH;        // [2L][LDH]
    float* dSs = KRs + 2 * L * LDH;    // [L][LS]
    float* Pds = dSs + L * LS;         // [L][LS]
    float* Drow = Pds + L * LS;        // [L]
    float* Lrow = Drow + L;            // [L]
    const int D = n_head * DH;
    const int h = blockIdx.y, hc = h * DH;
    const int lane = threadIdx.x;
    const int myc = lane % C;          // this lane's 4-column chunk in every stage-B item (64 % C == 0)
    const float4 rw4 = *reinterpret_cast<const float4*>(r_w_bias + hc + 4 * myc);
    const float4 rr4 = *reinterpret_cast<const float4*>(r_r_bias + hc + 4 * myc);
    const float4 del4 = make_float4(rr4.x - rw4.x, rr4.y - rw4.y, rr4.z - rw4.z, rr4.w - rw4.w);
    float4 acc_rw = make_float4(0.f, 0.f, 0.f, 0.f), acc_rr = acc_rw;
    float4 gkr[PAIRS_NKR];
#pragma unroll
    for (int u = 0; u < PAIRS_NKR; ++u) gkr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int n_kr_items = 2 * L * C;

    auto load_kr = [&](long boff) {
        for (int idx = lane; idx < 2 * L * C; idx += 64) {
            const int r = idx / C, c4 = idx % C;
            *reinterpret_cast<float4*>(KRs + r * LDH + 4 * c4) =
                *reinterpret_cast<const float4*>(kr + boff + (long)r * D + hc + 4 * c4);
        }
    };
    if (kr_bstride == 0) load_kr(0);

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        if (kr_bstride > 0) load_kr((long)b * kr_bstride);
        for (int idx = lane; idx < L * C; idx += 64) {
            const int r = idx / C, c4 = idx % C;     // c4 == myc
            const long g = ((long)b * L + r) * D + hc + 4 * c4;
            float4 t = *reinterpret_cast<const float4*>(q + g);
            t.x += rw4.x; t.y += rw4.y; t.z += rw4.z; t.w += rw4.w;
            *reinterpret_cast<float4*>(Qw + r * LDH + 4 * c4) = t;
            *reinterpret_cast<float4*>(Ks + r * LDH + 4 * c4) = *reinterpret_cast<const float4*>(k + g);
            *reinterpret_cast<float4*>(Vs + r * LDH + 4 * c4) = *reinterpret_cast<const float4*>(v + g);
            *reinterpret_cast<float4*>(dOs + r * LDH + 4 * c4) = *reinterpret_cast<const float4*>(dout + g);
        }
        if (lane < L) {
            const float* orow = out + ((long)b * L + lane) * D + hc;
            const float* grow = dout + ((long)b * L + lane) * D + hc;
            float dsum = 0.f;
#pragma unroll
            for (int d = 0; d < DH; d += 4) {
                const float4 a = *reinterpret_cast<const float4*>(orow + d);
                const float4 e = *reinterpret_cast<const float4*>(grow + d);
                dsum += a.x * e.x + a.y * e.y + a.z * e.z + a.w * e.w;
            }
            Drow[lane] = dsum;
            Lrow[lane] = lse[((long)b * n_head + h) * L + lane];
        }
        __syncthreads();
        // ---- stage A: (i, j) pairs
        const unsigned long long mbase = (unsigned long long)(b * n_head + h) * L;
        for (int p = lane; p < L * L; p += 64) {
            const int i = p / L, j = p - i * L;
            const float* qi = Qw + i * LDH;
            const float* kj = Ks + j * LDH;
            const float* krp = KRs + (j + L - i) * LDH;
            const float* gi = dOs + i * LDH;
            const float* vj = Vs + j * LDH;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < DH; d += 4) {
                const float4 a = *reinterpret_cast<const float4*>(qi + d);
                const float4 bk = *reinterpret_cast<const float4*>(kj + d);
                const float4 ck = *reinterpret_cast<const float4*>(krp + d);
                const float4 e = *reinterpret_cast<const float4*>(gi + d);
                const float4 f = *reinterpret_cast<const float4*>(vj + d);
                const float4 dl = *reinterpret_cast<const float4*>(r_r_bias + hc + d);
                const float4 dw = *reinterpret_cast<const float4*>(r_w_bias + hc + d);
                s += a.x * bk.x + a.y * bk.y + a.z * bk.z + a.w * bk.w;
                s += (a.x + dl.x - dw.x) * ck.x + (a.y + dl.y - dw.y) * ck.y + (a.z + dl.z - dw.z) * ck.z +
                     (a.w + dl.w - dw.w) * ck.w;
                dp += e.x * f.x + e.y * f.y + e.z * f.z + e.w * f.w;
            }
            const float pr = (key_len && j >= key_len[b] && j != i) ? 0.f : __expf(s * scale - Lrow[i]);
            const float msk = drop.p > 0.f ? drop_scale(drop, (mbase + i) * L + j) : 1.f;
            dSs[i * LS + j] = pr * (dp * msk - Drow[i]) * scale;
            Pds[i * LS + j] = pr * msk;
        }
        __syncthreads();
        // ---- stage B: d q (and the bias-gradient partials), rows i
        for (int item = lane; item < L * C; item += 64) {
            const int i = item / C;                  // chunk == myc
            float4 ga = make_float4(0.f, 0.f, 0.f, 0.f), gb = ga;
            for (int j = 0; j < L; ++j) {
                const float ds = dSs[i * LS + j];
                const float4 a = *reinterpret_cast<const float4*>(Ks + j * LDH + 4 * myc);
                const float4 c = *reinterpret_cast<const float4*>(KRs + (j + L - i) * LDH + 4 * myc);
                ga.x += ds * a.x; ga.y += ds * a.y; ga.z += ds * a.z; ga.w += ds * a.w;
                gb.x += ds * c.x; gb.y += ds * c.y; gb.z += ds * c.z; gb.w += ds * c.w;
            }
            *reinterpret_cast<float4*>(dq + ((long)b * L + i) * D + hc + 4 * myc) =
                make_float4(ga.x + gb.x, ga.y + gb.y, ga.z + gb.z, ga.w + gb.w);
            acc_rw.x += ga.x; acc_rw.y += ga.y; acc_rw.z += ga.z; acc_rw.w += ga.w;
            acc_rr.x += gb.x; acc_rr.y += gb.y; acc_rr.z += gb.z; acc_rr.w += gb.w;
        }
        // d k, d v : rows j
        for (int item = lane; item < L * C; item += 64) {
            const int j = item / C;
            float4 gk = make_float4(0.f, 0.f, 0.f, 0.f), gv = gk;
            for (int i = 0; i < L; ++i) {
                const float ds = dSs[i * LS + j];
                const float pd = Pds[i * LS + j];
                const float4 a = *reinterpret_cast<const float4*>(Qw + i * LDH + 4 * myc);
                const float4 e = *reinterpret_cast<const float4*>(dOs + i * LDH + 4 * myc);
                gk.x += ds * a.x; gk.y += ds * a.y; gk.z += ds * a.z; gk.w += ds * a.w;
                gv.x += pd * e.x; gv.y += pd * e.y; gv.z += pd * e.z; gv.w += pd * e.w;
            }
            *reinterpret_cast<float4*>(dk + ((long)b * L + j) * D + hc + 4 * myc) = gk;
            *reinterpret_cast<float4*>(dv + ((long)b * L + j) * D + hc + 4 * myc) = gv;
        }
        // d k_r : rows p in [0, 2L)
#pragma unroll
        for (int u = 0; u < PAIRS_NKR; ++u) {
            const int item = lane + 64 * u;
            if (item < n_kr_items) {
                const int pp = item / C;
                float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
                const int i_lo = max(0, L - pp), i_hi = min(L, 2 * L - pp);   // 0 <= pp - L + i < L
                for (int i = i_lo; i < i_hi; ++i) {
                    const float ds = dSs[i * LS + (pp - L + i)];
                    const float4 a = *reinterpret_cast<const float4*>(Qw + i * LDH + 4 * myc);
                    g.x += ds * (a.x + del4.x); g.y += ds * (a.y + del4.y);
                    g.z += ds * (a.z + del4.z); g.w += ds * (a.w + del4.w);
                }
                if (dkr_b) *reinterpret_cast<float4*>(dkr_b + ((long)b * 2 * L + pp) * D + hc + 4 * myc) = g;
                else { gkr[u].x += g.x; gkr[u].y += g.y; gkr[u].z += g.z; gkr[u].w += g.w; }
            }
        }
    }
    // ---- partial sums of this workgroup: row blockIdx.x of part[.][2L*D + 2D], this head's columns
    float* mypart = part + (long)blockIdx.x * (2 * L * D + 2 * D);
    if (!dkr_b) {
#pragma unroll
        for (int u = 0; u < PAIRS_NKR; ++u) {
            const int item = lane + 64 * u;
            if (item < n_kr_items)
                *reinterpret_cast<float4*>(mypart + (long)(item / C) * D + hc + 4 * myc) = gkr[u];
        }
    }
    // lanes sharing a chunk (lane % C) differ by multiples of C: butterfly over those
    for (int o = C; o < 64; o <<= 1) {
        acc_rw.x += __shfl_xor(acc_rw.x, o, 64); acc_rw.y += __shfl_xor(acc_rw.y, o, 64);
        acc_rw.z += __shfl_xor(acc_rw.z, o, 64); acc_rw.w += __shfl_xor(acc_rw.w, o, 64);
        acc_rr.x += __shfl_xor(acc_rr.x, o, 64); acc_rr.y += __shfl_xor(acc_rr.y, o, 64);
        acc_rr.z += __shfl_xor(acc_rr.z, o, 64); acc_rr.w += __shfl_xor(acc_rr.w, o, 64);
    }
    if (lane < C) {
        *reinterpret_cast<float4*>(mypart + 2 * L * D + hc + 4 * lane) = acc_rw;
        *reinterpret_cast<float4*>(mypart + 2 * L * D + D + hc + 4 * lane) = acc_rr;
    }
}

int t4r_reduce_partials_launch(hipStream_t st, const float* part, int nblocks, float* o0, int n0, int a0,
                               float* o1, int n1, int a1, float* o2, int n2, int a2);

static size_t attn_fwd_smem(int L, int D) { return (size_t)(4 * L) * (D + ATT_PAD) * sizeof(float); }
static size_t attn_bwd_smem(int L, int D, int n, int per_batch_kr = 0) {
    return ((size_t)((per_batch_kr ? 6 : 8) * L) * (D + ATT_PAD) + (size_t)2 * n * L * (L + 1) + 2 * D) * sizeof(float);
}

extern "C" int t4r_xlnet_attn_bwd_blocks(int B) { return B < 1024 ? B : 1024; }
static long attn_bwd_part_floats(int B, int L, int D, int n_head) {
    const int hpb = (D / n_head) >= 32 ? 4 : 8;
    return (long)t4r_xlnet_attn_bwd_blocks(B) * ((n_head + hpb - 1) / hpb) * (2L * L * D + 2L * D);
}
int t4r_xlnet_attn_mfma_ok(int L, int d_head);
// Which shapes leave the one-wave kernels for the general ones of xlnet_attn_long.hip: more than 64 positions, a head width
// without an instance, or (VALU kernels only: 32 < L <= 64, or d_head 8) a session whose K / V / k_r rows do not fit the LDS.
// dir: 0 forward, 1 backward with the shared k_r, 2 backward with per-session k_r, 3 either backward form
static bool attn_uses_long(int L, int D, int n_head, int dir) {
    const int d_head = n_head > 0 ? D / n_head : 0;
    if (L > 64 || !(d_head == 8 || d_head == 16 || d_head == 32)) return true;
    if (t4r_xlnet_attn_mfma_ok(L, d_head)) return false;            // (an experiment build may switch the MFMA kernels off: it
                                                                    // then fails loudly on the LDS check below, as before)
    const size_t lim = 160 * 1024;
    if (dir == 0) return attn_fwd_smem(L, D) > lim;
    if (dir == 1) return attn_bwd_smem(L, D, n_head, 0) > lim;
    if (dir == 2) return attn_bwd_smem(L, D, n_head, 1) > lim;
    return attn_bwd_smem(L, D, n_head, 0) > lim || attn_bwd_smem(L, D, n_head, 1) > lim;
}
// the general kernels keep delta [B, n_head, L] behind the partial rows
extern "C" long t4r_xlnet_attn_bwd_ws_floats(int B, int L, int D, int n_head) {
    return attn_bwd_part_floats(B, L, D, n_head) + (attn_uses_long(L, D, n_head, 3) ? (long)B * n_head * L : 0L);
}
int t4r_xlnet_attn_long_ok(int L, int d_head);
int t4r_xlnet_attn_long_fwd(hipStream_t st, const float* q, const float* k, const float* v, const float* kr, const float* rw,
                            const float* rr, float* out, float* lse, int B, int L, int n_head, int d_head, float scale,
                            long kr_bstride, DropCfg drop, const int* key_len);
int t4r_xlnet_attn_long_bwd(hipStream_t st, const float* q, const float* k, const float* v, const float* kr, const float* rw,
                            const float* rr, const float* out, const float* lse, const float* dout, float* dq, float* dk,
                            float* dv, float* part, float* delta, float* dkr, float* d_rw, float* d_rr, int B, int L, int n_head,
                            int d_head, float scale, long kr_bstride, DropCfg drop, const int* key_len);

// MFMA kernels for L <= 32, d_head 16 / 32 (xlnet_attn_mfma.hip); T4R_ATTN_MFMA=0 keeps the VALU kernels
int t4r_xlnet_attn_mfma_ok(int L, int d_head);
int t4r_xlnet_attn_mfma_fwd(hipStream_t st, const float* q, const float* k, const float* v, const float* kr,
                            const float* rw, const float* rr, float* out, float* lse, int B, int L, int n_head,
                            int d_head, float scale, long kr_bstride, DropCfg drop, const int* key_len);
int t4r_xlnet_attn_mfma_bwd(hipStream_t st, const float* q, const float* k, const float* v, const float* kr,
                            const float* rw, const float* rr, const float* lse, const float* dout, float* dq,
                            float* dk, float* dv, float* part, float* dkr, float* d_rw, float* d_rr, int B, int L,
                            int n_head, int d_head, float scale, long kr_bstride, DropCfg drop, const int* key_len);
#ifdef T4R_EXPERIMENTAL
// the fp32-MFMA core with the permuted k-slots (tools/experimental: phase 3 of the one-kernel backward as its own launch;
// measured slower than xlnet_attn_mfma_bwd_kernel, not in the product library)
int t4r_xlnet_attn_core16_ok(int L, int D, int n_head);
int t4r_xlnet_attn_core16_bwd(hipStream_t st, const float* qkv, const float* kr, const float* rw, const float* rr,
                              const float* lse, const float* dout, float* dqkv, float* part, float* dkr, float* d_rw,
                              float* d_rr, int B, int L, int n_head, int d_head, float scale, long kr_bstride, DropCfg drop,
                              const int* key_len);
#endif
static bool use_mfma(int L, int d_head) {
    static int en = -1;
    if (en < 0) { const char* e = t4r_exp_getenv("T4R_ATTN_MFMA"); en = e ? atoi(e) : 1; }
    return en && t4r_xlnet_attn_mfma_ok(L, d_head);
}

template <int DH>
static int attn_fwd_launch(hipStream_t st, const float* q, const float* k, const float* v,
                           const float* kr, const float* rw, const float* rr, float* out, float* lse,
                           int B, int L, int n_head, float scale, long kr_bstride, DropCfg drop, const int* key_len) {
    const int D = n_head * DH;
    const size_t smem = attn_fwd_smem(L, D);
    static T4rLdsAttr attr;
    t4r_ensure_dynamic_lds((const void*)xlnet_attn_fwd_kernel<DH>, smem, attr);
    constexpr int HPB = HeadsPerBlock<DH>::v;
    const int waves = n_head < HPB ? n_head : HPB;
    hipLaunchKernelGGL(xlnet_attn_fwd_kernel<DH>, dim3(B, (n_head + HPB - 1) / HPB), dim3(64 * waves), smem, st, q, k, v, kr, rw,
                       rr, out, lse, B, L, n_head, scale, kr_bstride, drop, key_len);
    T4R_LAUNCH_CHECK();
    return 0;
}

// kr_per_batch: k_r is [B, 2L, D] (one set of positional keys per session: pos_emb dropout) instead
// of [2L, D].  drop_p: attention-probability dropout (mask index ((b*n+h)*L+i)*L+j).
extern "C" int t4r_xlnet_attn_fwd(void* stream, const float* q, const float* k, const float* v,
                                  const float* k_r, const float* r_w_bias, const float* r_r_bias,
                                  float* out, float* lse, int B, int L, int n_head, int d_head,
                                  int kr_per_batch, float drop_p, unsigned long long seed,
                                  unsigned long long ctr_hi, const int* key_len) {
    if (B == 0) return 0;
    T4R_CHECK_ARG(L >= 1, "xlnet_attn: L must be at least 1");
    const int D = n_head * d_head;
    const float scale = 1.0f / sqrtf((float)d_head);
    hipStream_t st = (hipStream_t)stream;
    const long bs = kr_per_batch ? 2L * L * D : 0;
    const DropCfg dc = make_drop(drop_p, seed, ctr_hi);
    if (attn_uses_long(L, D, n_head, 0)) {       // beyond one wave per row block, a head width the one-wave kernels have no
        // instance for, or rows that do not fit the LDS: the general kernels (any L, d_head up to 256)
        T4R_CHECK_ARG(t4r_xlnet_attn_long_ok(L, d_head), "xlnet_attn: d_head must be at most 256");
        return t4r_xlnet_attn_long_fwd(st, q, k, v, k_r, r_w_bias, r_r_bias, out, lse, B, L, n_head, d_head, scale, bs, dc, key_len);
    }
    if (use_mfma(L, d_head))
        return t4r_xlnet_attn_mfma_fwd(st, q, k, v, k_r, r_w_bias, r_r_bias, out, lse, B, L, n_head, d_head, scale,
                                       bs, dc, key_len);
    T4R_CHECK_ARG(attn_fwd_smem(L, D) <= 160 * 1024, "xlnet_attn: L*d_model too large for LDS");
    switch (d_head) {
        case 8: return attn_fwd_launch<8>(st, q, k, v, k_r, r_w_bias, r_r_bias, out, lse, B, L, n_head, scale, bs, dc, key_len);
        case 16: return attn_fwd_launch<16>(st, q, k, v, k_r, r_w_bias, r_r_bias, out, lse, B, L, n_head, scale, bs, dc, key_len);
        case 32: return attn_fwd_launch<32>(st, q, k, v, k_r, r_w_bias, r_r_bias, out, lse, B, L, n_head, scale, bs, dc, key_len);
        case 64: return attn_fwd_launch<64>(st, q, k, v, k_r, r_w_bias, r_r_bias, out, lse, B, L, n_head, scale, bs, dc, key_len);
    }
    t4r_set_error("xlnet_attn: d_head must be 8, 16, 32 or 64");
    return -1;
}

template <int DH>
static int attn_bwd_launch(hipStream_t st, const float* q, const float* k, const float* v,
                           const float* kr, const float* rw, const float* rr, const float* out,
                           const float* lse, const float* dout, float* dq, float* dk, float* dv,
                           float* part, float* dkr, float* d_rw, float* d_rr, int B, int L, int n_head,
                           float scale, long kr_bstride, DropCfg drop, const int* key_len) {
    const int D = n_head * DH;
    if (L <= 32 && (DH == 16 || DH == 32 || DH == 8) && 2 * L * (DH / 4) <= 64 * PAIRS_NKR) {
        const size_t sm2 = ((size_t)6 * L * (DH + 4) + 2 * L * (L + 1) + 2 * L) * sizeof(float);
        const int gx = B < 512 ? B : 512;
        hipLaunchKernelGGL(xlnet_attn_bwd_pairs_kernel<DH>, dim3(gx, n_head), dim3(64), sm2, st, q, k, v, kr, rw, rr,
                           out, lse, dout, dq, dk, dv, part, kr_bstride > 0 ? dkr : nullptr, B, L, n_head, scale,
                           kr_bstride, drop, key_len);
        T4R_LAUNCH_CHECK();
        return t4r_reduce_partials_launch(st, part, gx, kr_bstride > 0 ? nullptr : dkr, 2 * L * D, 0, d_rw, D, 1,
                                          d_rr, D, 1);
    }
    const size_t smem = attn_bwd_smem(L, D, n_head, kr_bstride > 0);
    static T4rLdsAttr attr;
    t4r_ensure_dynamic_lds((const void*)xlnet_attn_bwd_kernel<DH>, smem, attr);
    constexpr int HPB = HeadsPerBlock<DH>::v;
    const int waves = n_head < HPB ? n_head : HPB;
    const int hg = (n_head + HPB - 1) / HPB;
    const int nblocks = t4r_xlnet_attn_bwd_blocks(B);
    hipLaunchKernelGGL(xlnet_attn_bwd_kernel<DH>, dim3(nblocks, hg), dim3(64 * waves), smem, st, q, k, v, kr,
                       rw, rr, out, lse, dout, dq, dk, dv, part, kr_bstride > 0 ? dkr : nullptr, B, L, n_head,
                       scale, kr_bstride, drop, key_len);
    T4R_LAUNCH_CHECK();
    // d k_r overwritten (shared k_r: summed over sessions here), bias gradients accumulated
    return t4r_reduce_partials_launch(st, part, nblocks * hg, kr_bstride > 0 ? nullptr : dkr, 2 * L * D, 0,
                                      d_rw, D, 1, d_rr, D, 1);
}

// d_rw / d_rr are ACCUMULATED into (parameter gradients); dq/dk/dv/dk_r are overwritten.
extern "C" int t4r_xlnet_attn_bwd(void* stream, const float* q, const float* k, const float* v,
                                  const float* k_r, const float* r_w_bias, const float* r_r_bias,
                                  const float* out, const float* lse, const float* dout, float* dq,
                                  float* dk, float* dv, float* dk_r, float* d_r_w_bias,
                                  float* d_r_r_bias, float* workspace, int B, int L, int n_head,
                                  int d_head, int kr_per_batch, float drop_p, unsigned long long seed,
                                  unsigned long long ctr_hi, const int* key_len) {
    if (B == 0) return 0;
    T4R_CHECK_ARG(L >= 1, "xlnet_attn: L must be at least 1");
    const int D = n_head * d_head;
    const float scale = 1.0f / sqrtf((float)d_head);
    hipStream_t st = (hipStream_t)stream;
    const long bs = kr_per_batch ? 2L * L * D : 0;
    const DropCfg dc = make_drop(drop_p, seed, ctr_hi);
    if (attn_uses_long(L, D, n_head, kr_per_batch ? 2 : 1)) {
        T4R_CHECK_ARG(t4r_xlnet_attn_long_ok(L, d_head), "xlnet_attn_bwd: d_head must be at most 256");
        T4R_CHECK_ARG(out != nullptr, "xlnet_attn_bwd: the forward output is needed by the general kernels (L > 64 or d_head not 8 / 16 / 32)");
        return t4r_xlnet_attn_long_bwd(st, q, k, v, k_r, r_w_bias, r_r_bias, out, lse, dout, dq, dk, dv, workspace,
                                       workspace + attn_bwd_part_floats(B, L, D, n_head), dk_r, d_r_w_bias, d_r_r_bias, B, L,
                                       n_head, d_head, scale, bs, dc, key_len);
    }
#ifdef T4R_EXPERIMENTAL
    {
        const long TD = (long)B * L * D;
        if (k == q + TD && v == q + 2 * TD && dk == dq + TD && dv == dq + 2 * TD && t4r_xlnet_attn_core16_ok(L, D, n_head))
            return t4r_xlnet_attn_core16_bwd(st, q, k_r, r_w_bias, r_r_bias, lse, dout, dq, workspace, dk_r, d_r_w_bias,
                                             d_r_r_bias, B, L, n_head, d_head, scale, bs, dc, key_len);
    }
#endif
    if (use_mfma(L, d_head))
        return t4r_xlnet_attn_mfma_bwd(st, q, k, v, k_r, r_w_bias, r_r_bias, lse, dout, dq, dk, dv, workspace, dk_r,
                                       d_r_w_bias, d_r_r_bias, B, L, n_head, d_head, scale, bs, dc, key_len);
    T4R_CHECK_ARG(attn_bwd_smem(L, D, n_head) <= 160 * 1024, "xlnet_attn_bwd: L*d_model too large for LDS");
    switch (d_head) {
        case 8: return attn_bwd_launch<8>(st, q, k, v, k_r, r_w_bias, r_r_bias, out, lse, dout, dq, dk, dv, workspace, dk_r, d_r_w_bias, d_r_r_bias, B, L, n_head, scale, bs, dc, key_len);
        case 16: return attn_bwd_launch<16>(st, q, k, v, k_r, r_w_bias, r_r_bias, out, lse, dout, dq, dk, dv, workspace, dk_r, d_r_w_bias, d_r_r_bias, B, L, n_head, scale, bs, dc, key_len);
        case 32: return attn_bwd_launch<32>(st, q, k, v, k_r, r_w_bias, r_r_bias, out, lse, dout, dq, dk, dv, workspace, dk_r, d_r_w_bias, d_r_r_bias, B, L, n_head, scale, bs, dc, key_len);
    }
    t4r_set_error("xlnet_attn_bwd: d_head must be 8, 16 or 32");
    return -1;
}
