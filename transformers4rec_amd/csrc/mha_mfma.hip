// Scaled-dot-product attention core (GPT-2 causal / BERT bidirectional) on the fp32 matrix cores,
// for L <= 128 and d_head in {32, 64}: ONE WAVE per (session, head), the [L x L] score matrix is
// walked in 32 x 32 tiles and every contraction is a chain of v_mfma_f32_32x32x2_f32.
//
// Same arithmetic as the VALU kernels in mha.hip (which stay as the fallback for d_head 16 and as
// the A/B reference, T4R_MHA_MFMA=0), restating HF gpt2/modeling_gpt2.py eager_attention_forward
// :54-72 and HF bert BertSelfAttention as instantiated by transformers4rec/config/transformer.py
// :218-260 / :493-534:   out = dropout(softmax(q k^T / sqrt(dh) [+ causal])) v   and its autograd.
//
// Forward, per 32-row query tile: the scores of the whole row (up to four key tiles) are brought to
// the row layout (lane (i, kh) owns columns kh*16..kh*16+15 of every key tile) through one
// [32][33] LDS exchange buffer, the softmax runs in registers, and the normalised (dropped)
// probabilities are directly the A operand of P V.
// Backward, key tile outer / query tile inner: d k_j and d v_j accumulate in MFMA accumulators over
// the query tiles; d q_i is accumulated in place in global memory (every (row, column) is owned by
// one lane of one wave, first key tile stores, later ones read-modify-write: deterministic, L2
// resident).  P is recomputed from the saved log-sum-exp, D_i = dO_i . O_i parks in LDS.
// Measured before this kernel (BERT-like B 256, L 100, 8 heads x 64): mha_bwd 2.0 ms, mha_fwd
// 0.47 ms per layer.
#include "t4r_common.h"

#include "mfma_frag.h"

// column fragment over rows row0 .. row0+15 of a row-major matrix with row stride ld (k = row index);
// rows >= nrows are clamped (the other operand is zero there)
__device__ __forceinline__ Frag<16> col_frag_rows(const float* base, int ld, int row0, int nrows, int col) {
    Frag<16> f;
#pragma unroll
    for (int s = 0; s < 16; ++s) f.v[s] = base[min(row0 + s, nrows - 1) * ld + col];
    return f;
}

#define MHA_LDS_WAIT()                          \
    __builtin_amdgcn_s_waitcnt(0xc07f);         \
    __builtin_amdgcn_wave_barrier()

// ------------------------------------------------------------------------------------------ forward
template <int DH>
__global__ __launch_bounds__(512) void mha_mfma_fwd_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ld,
    float* __restrict__ out, int ld_out, float* __restrict__ lse, int B, int L, int n_head, float scale,
    int causal, DropCfg drop, const int* key_len) {
    __shared__ __attribute__((aligned(16))) float Sm[32 * XM_SP];
    const int lane = threadIdx.x;
    const int h = blockIdx.y, hc = h * DH;
    const int nT = (L + 31) >> 5;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const int klen = key_len ? max(1, min(L, key_len[b])) : L;   // opt-in padding mask: keys >= klen masked
        const float* qb = q + (long)b * L * ld + hc;
        const float* kb = k + (long)b * L * ld + hc;
        const float* vb = v + (long)b * L * ld + hc;
        float* ob = out + (long)b * L * ld_out + hc;
        float* lb = lse + ((long)b * n_head + h) * L;
        for (int qi = 0; qi < nT; ++qi) {
            int c = lane & 31, kh = lane >> 5;
            asm volatile("" : "+v"(c), "+v"(kh));
            const int i = qi * 32 + c;
            const Frag<DH / 2> qf = row_frag<DH>(qb, min(i, L - 1) * ld + kh * (DH / 2));
            Frag<16> s[4];
#pragma unroll
            for (int kj = 0; kj < 4; ++kj) {
                const bool live = kj < nT && !(causal && kj > qi);      // wave uniform
                if (live) {
                    const Frag<DH / 2> kf = row_frag<DH>(kb, min(kj * 32 + c, L - 1) * ld + kh * (DH / 2));
                    f32x16 acc = zero16();
                    mfma_chain(acc, qf, kf);
#pragma unroll
                    for (int r = 0; r < 16; ++r) Sm[xm_row(r, kh) * XM_SP + c] = acc[r];
                    MHA_LDS_WAIT();
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const int j = kj * 32 + kh * 16 + t;
                        const float x = Sm[c * XM_SP + kh * 16 + t] * scale;
                        s[kj].v[t] = (j < klen && (!causal || j <= i)) ? x : -INFINITY;
                    }
                    __builtin_amdgcn_wave_barrier();
                } else {
#pragma unroll
                    for (int t = 0; t < 16; ++t) s[kj].v[t] = -INFINITY;
                }
            }
            float m = -INFINITY;
#pragma unroll
            for (int kj = 0; kj < 4; ++kj)
#pragma unroll
                for (int t = 0; t < 16; ++t) m = fmaxf(m, s[kj].v[t]);
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int kj = 0; kj < 4; ++kj)
#pragma unroll
                for (int t = 0; t < 16; ++t) { s[kj].v[t] = __expf(s[kj].v[t] - m); sum += s[kj].v[t]; }
            sum += __shfl_xor(sum, 32, 64);
            const float inv = 1.f / sum;
            if (kh == 0 && i < L) lb[i] = m + __logf(sum);
            const unsigned long long mbase = ((unsigned long long)(b * n_head + h) * L + min(i, L - 1)) * L;
#pragma unroll
            for (int kj = 0; kj < 4; ++kj) {
#pragma unroll
                for (int t = 0; t < 16; ++t) s[kj].v[t] *= inv;
                if (drop.p > 0.f && kj < nT && !(causal && kj > qi)) {
                    float msk[16];
                    const int j0 = kj * 32 + kh * 16;
                    drop_scale_run16(drop, mbase + j0, L - j0, (L & 3) == 0, msk);
#pragma unroll
                    for (int t = 0; t < 16; ++t) s[kj].v[t] *= msk[t];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // out_i = sum_j P~[i][j] v_j : A = row layout, B = column fragment of V (k = j)
#pragma unroll
            for (int dt = 0; dt < DH / 32; ++dt) {
                f32x16 o = zero16();
#pragma unroll
                for (int kj = 0; kj < 4; ++kj) {
                    if (kj < nT && !(causal && kj > qi)) {
                        const Frag<16> vf = col_frag_rows(vb, ld, kj * 32 + kh * 16, L, dt * 32 + c);
                        mfma_chain(o, s[kj], vf);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = qi * 32 + xm_row(r, kh);
                    if (row < L) ob[row * ld_out + dt * 32 + c] = o[r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ backward
template <int DH>
__global__ __launch_bounds__(512) void mha_mfma_bwd_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ld,
    const float* __restrict__ out, const float* __restrict__ dout, int ld_out, const float* __restrict__ lse,
    float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv, int ld_d, int B, int L, int n_head,
    float scale, int causal, DropCfg drop, const int* key_len) {
    __shared__ __attribute__((aligned(16))) float Sm[32 * XM_SP];
    __shared__ __attribute__((aligned(16))) float Pm[32 * XM_SP];   // dropped probabilities, parked for d v
    __shared__ float Dr[128];                                       // D_i = dO_i . O_i
    const int lane = threadIdx.x;
    const int h = blockIdx.y, hc = h * DH;
    const int nT = (L + 31) >> 5;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const int klen = key_len ? max(1, min(L, key_len[b])) : L;
        const float* qb = q + (long)b * L * ld + hc;
        const float* kb = k + (long)b * L * ld + hc;
        const float* vb = v + (long)b * L * ld + hc;
        const float* ob = out + (long)b * L * ld_out + hc;
        const float* gb = dout + (long)b * L * ld_out + hc;
        float* dqb = dq + (long)b * L * ld_d + hc;
        float* dkb = dk + (long)b * L * ld_d + hc;
        float* dvb = dv + (long)b * L * ld_d + hc;
        const float* lb = lse + ((long)b * n_head + h) * L;
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < L; i += 64) {
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < DH; e += 4) {
                const float4 a = *reinterpret_cast<const float4*>(gb + i * ld_out + e);
                const float4 o = *reinterpret_cast<const float4*>(ob + i * ld_out + e);
                d += a.x * o.x + a.y * o.y + a.z * o.z + a.w * o.w;
            }
            Dr[i] = d;
        }
        MHA_LDS_WAIT();
        for (int kj = 0; kj < nT; ++kj) {
            f32x16 gk[DH / 32], gv[DH / 32];
#pragma unroll
            for (int dt = 0; dt < DH / 32; ++dt) { gk[dt] = zero16(); gv[dt] = zero16(); }
            const int j0 = kj * 32;
            for (int qi = causal ? kj : 0; qi < nT; ++qi) {
                int c = lane & 31, kh = lane >> 5;
                asm volatile("" : "+v"(c), "+v"(kh));
                const int i = qi * 32 + c, ic = min(i, L - 1);
                const int jc = min(j0 + c, L - 1);
                // S = Q_i K_j^T (accumulator layout) -> row layout through Sm
                {
                    const Frag<DH / 2> qf = row_frag<DH>(qb, ic * ld + kh * (DH / 2));
                    const Frag<DH / 2> kf = row_frag<DH>(kb, jc * ld + kh * (DH / 2));
                    f32x16 acc = zero16();
                    mfma_chain(acc, qf, kf);
#pragma unroll
                    for (int r = 0; r < 16; ++r) Sm[xm_row(r, kh) * XM_SP + c] = acc[r];
                }
                MHA_LDS_WAIT();
                const float lrow = lb[ic];
                Frag<16> P;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int j = j0 + kh * 16 + t;
                    const bool ok = i < L && j < klen && (!causal || j <= i);
                    P.v[t] = ok ? __expf(Sm[c * XM_SP + kh * 16 + t] * scale - lrow) : 0.f;
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // dP = dO_i V_j^T
                {
                    const Frag<DH / 2> gf = row_frag<DH>(gb, ic * ld_out + kh * (DH / 2));
                    const Frag<DH / 2> vf = row_frag<DH>(vb, jc * ld + kh * (DH / 2));
                    f32x16 acc = zero16();
                    mfma_chain(acc, gf, vf);
#pragma unroll
                    for (int r = 0; r < 16; ++r) Sm[xm_row(r, kh) * XM_SP + c] = acc[r];
                }
                MHA_LDS_WAIT();
                Frag<16> dS;
                {
                    const float drow = Dr[ic];
                    const unsigned long long mbase = ((unsigned long long)(b * n_head + h) * L + ic) * L;
                    float msk[16];
                    if (drop.p > 0.f) {
                        drop_scale_run16(drop, mbase + j0 + kh * 16, L - j0 - kh * 16, (L & 3) == 0, msk);
                    } else {
#pragma unroll
                        for (int t = 0; t < 16; ++t) msk[t] = 1.f;
                    }
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const float dpm = Sm[c * XM_SP + kh * 16 + t] * msk[t];
                        Pm[c * XM_SP + kh * 16 + t] = P.v[t] * msk[t];
                        dS.v[t] = P.v[t] * (dpm - drow) * scale;
                    }
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < 16; ++t) Sm[c * XM_SP + kh * 16 + t] = dS.v[t];
                MHA_LDS_WAIT();
                __builtin_amdgcn_sched_barrier(0);
                // d q_i += dS K_j : A = row layout, B = column fragment of K (k = j)
#pragma unroll
                for (int dt = 0; dt < DH / 32; ++dt) {
                    const Frag<16> kf = col_frag_rows(kb, ld, j0 + kh * 16, L, dt * 32 + c);
                    f32x16 acc = zero16();
                    mfma_chain(acc, dS, kf);
                    const bool first = kj == 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = qi * 32 + xm_row(r, kh);
                        if (row < L) {
                            float* p = dqb + row * ld_d + dt * 32 + c;
                            if (first) *p = acc[r];
                            else *p += acc[r];
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // d k_j += dS^T Q_i ; d v_j += P~^T dO_i : A read by columns from LDS, B = column fragment (k = i)
                {
                    Frag<16> at;
#pragma unroll
                    for (int t = 0; t < 16; ++t) at.v[t] = Sm[(kh * 16 + t) * XM_SP + c];
#pragma unroll
                    for (int dt = 0; dt < DH / 32; ++dt) {
                        const Frag<16> qf = col_frag_rows(qb, ld, qi * 32 + kh * 16, L, dt * 32 + c);
                        mfma_chain(gk[dt], at, qf);
                    }
#pragma unroll
                    for (int t = 0; t < 16; ++t) at.v[t] = Pm[(kh * 16 + t) * XM_SP + c];
#pragma unroll
                    for (int dt = 0; dt < DH / 32; ++dt) {
                        const Frag<16> gf = col_frag_rows(gb, ld_out, qi * 32 + kh * 16, L, dt * 32 + c);
                        mfma_chain(gv[dt], at, gf);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                int c = lane & 31, kh = lane >> 5;
                asm volatile("" : "+v"(c), "+v"(kh));
#pragma unroll
                for (int dt = 0; dt < DH / 32; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int j = j0 + xm_row(r, kh);
                        if (j < L) {
                            dkb[j * ld_d + dt * 32 + c] = gk[dt][r];
                            dvb[j * ld_d + dt * 32 + c] = gv[dt][r];
                        }
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ host
bool t4r_mha_mfma_ok(int L, int d_head, long ld, long ld_out, long ld_d) {
    static const int on = [] {
        const char* e = t4r_exp_getenv("T4R_MHA_MFMA");
        return e ? atoi(e) : 1;
    }();
    const long lim = 0x7fffffffL / 160;     // 32-bit per-session offsets
    return on && L >= 1 && L <= 128 && (d_head == 32 || d_head == 64) && ld < lim && ld_out < lim && ld_d < lim;
}

static int mha_mfma_blocks(int B) { return B < 8192 ? B : 8192; }

int t4r_mha_mfma_fwd(hipStream_t st, const float* q, const float* k, const float* v, long ld, float* out,
                     long ld_out, float* lse, int B, int L, int n_head, int d_head, float scale, int causal,
                     DropCfg drop, const int* key_len) {
    const dim3 grid(mha_mfma_blocks(B), n_head), block(64);
    if (d_head == 64)
        hipLaunchKernelGGL(mha_mfma_fwd_kernel<64>, grid, block, 0, st, q, k, v, (int)ld, out, (int)ld_out, lse, B, L,
                           n_head, scale, causal, drop, key_len);
    else
        hipLaunchKernelGGL(mha_mfma_fwd_kernel<32>, grid, block, 0, st, q, k, v, (int)ld, out, (int)ld_out, lse, B, L,
                           n_head, scale, causal, drop, key_len);
    T4R_LAUNCH_CHECK();
    return 0;
}

int t4r_mha_mfma_bwd(hipStream_t st, const float* q, const float* k, const float* v, long ld, const float* out,
                     const float* dout, long ld_out, const float* lse, float* dq, float* dk, float* dv, long ld_d,
                     int B, int L, int n_head, int d_head, float scale, int causal, DropCfg drop, const int* key_len) {
    const dim3 grid(mha_mfma_blocks(B), n_head), block(64);
    if (d_head == 64)
        hipLaunchKernelGGL(mha_mfma_bwd_kernel<64>, grid, block, 0, st, q, k, v, (int)ld, out, dout, (int)ld_out, lse,
                           dq, dk, dv, (int)ld_d, B, L, n_head, scale, causal, drop, key_len);
    else
        hipLaunchKernelGGL(mha_mfma_bwd_kernel<32>, grid, block, 0, st, q, k, v, (int)ld, out, dout, (int)ld_out, lse,
                           dq, dk, dv, (int)ld_d, B, L, n_head, scale, causal, drop, key_len);
    T4R_LAUNCH_CHECK();
    return 0;
}
