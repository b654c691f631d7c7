// XLNet relative-position attention core on the fp32 matrix cores, for L <= 32 and
// d_head in {16, 32}: ONE WAVE per (session, head), every contraction of the forward and the
// backward is a chain of v_mfma_f32_32x32x2_f32 on one 32x32 tile (the sequence is padded to 32).
//
// Reference behaviour restated (HF transformers/models/xlnet/modeling_xlnet.py, XLNetRelativeAttention):
//   ac = (q + r_w_bias) . k                      :rel_attn_core, einsum("ibnd,jbnd->bnij")
//   bd = rel_shift((q + r_r_bias) . k_r)         bd[i, j] = raw[i, j + L - i]  (:rel_shift_bnij)
//   p  = softmax_j((ac + bd) * d_head^-0.5)      no attention mask on this path
//   out = dropout(p) @ v                         (self.dropout on the probabilities)
// and its autograd.  The VALU/LDS kernels in xlnet_attn.hip stay as the general fallback
// (L up to 64, other head widths); measured at C2 (B 1024, 4 heads x 32, L 20): forward
// 40.6 us, backward 115 us per layer with those.
//
// MFMA operand conventions (lane l: c = l & 31, kh = l >> 5):
//   D[i][j] += sum_ks A[i][ks] * B[ks][j]; the lane supplies A[i = c][ks = kh], B[ks = kh][j = c]
//   and holds D[(r & 3) + 8 * (r >> 2) + 4 * kh][c] in accumulator register r.
//   The k-slots are permuted: step s of a contraction of length K takes k = kh * K/2 + s, so
//   * an operand whose k runs along a row of a row-major matrix is a CONTIGUOUS run per lane
//     ("row fragment": float4 loads straight from HBM/L2, no LDS staging), and
//   * a [32 x 32] matrix held one row per lane pair (lane (i, kh) owns columns kh*16 .. kh*16+15:
//     the "row layout" in which the softmax runs) IS the A operand of a contraction over its columns.
//   Operands whose k is the ROW index of a row-major matrix ("column fragment") are 128-byte
//   coalesced scalar loads.  Only layout changes go through LDS (accumulator layout -> row layout,
//   transposes, the rel_shift gather / scatter): ~12.5 KB per wave.
#include "t4r_common.h"

#include "mfma_frag.h"

// Register budgets (launch bound / 64 threads launched): the backward keeps 17 KB of LDS per wave, so at
// most 9 waves fit a CU anyway; 512 (<= 256 VGPRs, 146 used) measured 67 us vs 80 us at 1024 (<= 128
// VGPRs), 69 us without the phase fences, 66-76 us at 256 (C2 shape, tools/attn_bench.py).
// With the operand fragments requested one phase ahead: forward 21.0 us at 512 (167 VGPRs, was 26.2 us
// at 1024 with loads at their uses), backward 60 us at 512 (213 VGPRs; 256 gives 60 / 72 us for
// per-session / shared k_r).
#ifndef T4R_ATTN_BWD_BOUNDS
#define T4R_ATTN_BWD_BOUNDS 512
#endif
#ifndef T4R_ATTN_FWD_BOUNDS
#define T4R_ATTN_FWD_BOUNDS 512
#endif
#ifndef T4R_ATTN_FWD_FENCE
#define T4R_ATTN_FWD_FENCE 1
#endif
#ifndef T4R_ATTN_FENCE
#define T4R_ATTN_FENCE 1
#endif
#if T4R_ATTN_FENCE
#define ATTN_PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define ATTN_PHASE_FENCE()
#endif

// scores of one (session, head) in ROW LAYOUT: lane (i = c, kh) gets s[t] = (ac + bd)[i][kh*16 + t] * scale
// (garbage where i >= L or j >= L: the caller masks).  Uses Sm [32][XM_SP] and Rm [32][XM_RP].
// The operand fragments are loaded by the caller, all at once at the top of a session (with the operands
// of its later phases): the wave then pays the memory latency once instead of once per contraction.
template <int DH>
struct ScoreFrags { Frag<DH / 2> q, k, kr0, kr1; };
template <int DH>
__device__ __forceinline__ ScoreFrags<DH> load_score_frags(const float* qb, const float* kb, const float* krb, int D,
                                                           int hc, int L, int c, int kh) {
    ScoreFrags<DH> f;
    const int roff = min(c, L - 1) * D + hc + kh * (DH / 2);
    f.q = row_frag<DH>(qb, roff);
    f.k = row_frag<DH>(kb, roff);                                                   // B: lane column j = c
    f.kr0 = row_frag<DH>(krb, min(c, 2 * L - 1) * D + hc + kh * (DH / 2));          // B: lane column m = c
    f.kr1 = row_frag<DH>(krb, min(c + 32, 2 * L - 1) * D + hc + kh * (DH / 2));     //               m = c + 32
    return f;
}
template <int DH>
__device__ __forceinline__ void scores_row_layout(const ScoreFrags<DH>& f, int L, const Frag<DH / 2>& rw,
                                                  const Frag<DH / 2>& del, float scale, float* Sm, float* Rm,
                                                  int c, int kh, float (&s)[16]) {
    const int ic = min(c, L - 1);
    Frag<DH / 2> qw = f.q;
#pragma unroll
    for (int t = 0; t < DH / 2; ++t) qw.v[t] += rw.v[t];
    {
        f32x16 ac = zero16();
        mfma_chain(ac, qw, f.k);
#pragma unroll
        for (int r = 0; r < 16; ++r) Sm[xm_row(r, kh) * XM_SP + c] = ac[r];
    }
#pragma unroll
    for (int t = 0; t < DH / 2; ++t) qw.v[t] += del.v[t];                       // q + r_r_bias
    {
        f32x16 raw = zero16();
        mfma_chain(raw, qw, f.kr0);
#pragma unroll
        for (int r = 0; r < 16; ++r) Rm[xm_row(r, kh) * XM_RP + c] = raw[r];
    }
    {
        f32x16 raw = zero16();
        mfma_chain(raw, qw, f.kr1);
#pragma unroll
        for (int r = 0; r < 16; ++r) Rm[xm_row(r, kh) * XM_RP + c + 32] = raw[r];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes are done (single-wave workgroup)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int j = kh * 16 + t;
        const int jj = min(j, L - 1);      // keeps the shifted index inside the row for masked columns
        s[t] = (Sm[c * XM_SP + j] + Rm[c * XM_RP + jj + L - ic]) * scale;
    }
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------ forward
template <int DH>
// launch bound 1024 although 64 threads are launched: it caps the register budget at 128 (4 waves/SIMD)
__global__ __launch_bounds__(T4R_ATTN_FWD_BOUNDS) void xlnet_attn_mfma_fwd_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ kr, const float* __restrict__ r_w_bias, const float* __restrict__ r_r_bias,
    float* __restrict__ out, float* __restrict__ lse, int B, int L, int n_head, float scale, long kr_bstride,
    DropCfg drop, const int* key_len) {
    __shared__ __attribute__((aligned(16))) float Sm[32 * XM_SP];
    __shared__ __attribute__((aligned(16))) float Rm[32 * XM_RP];
    const int lane = threadIdx.x;
    const int h = blockIdx.y, hc = h * DH, D = n_head * DH;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        // lane coordinates are laundered per session: everything derived from them (offsets, clamps)
        // is then recomputed where it is used instead of being hoisted out of the loop and kept live
        int c = lane & 31, kh = lane >> 5;
        asm volatile("" : "+v"(c), "+v"(kh));
        Frag<DH / 2> rw, del;
#pragma unroll
        for (int t = 0; t < DH / 2; ++t) {
            rw.v[t] = r_w_bias[hc + kh * (DH / 2) + t];
            del.v[t] = r_r_bias[hc + kh * (DH / 2) + t] - rw.v[t];
        }
        const long tok0 = (long)b * L;
        const float* qb = q + tok0 * D;
        const float* kb = k + tok0 * D;
        const float* vb = v + tok0 * D;
        float* ob = out + tok0 * D;
        const ScoreFrags<DH> sf = load_score_frags<DH>(qb, kb, kr + (long)b * kr_bstride, D, hc, L, c, kh);
        const Frag<16> vf = col_frag<16>(vb, hc + min(c, DH - 1), D, L, kh, 0.f);    // for P~ V, in flight early
        __builtin_amdgcn_sched_barrier(0);    // the loads are issued here; the waits sit at the uses
        float s[16];
        scores_row_layout<DH>(sf, L, rw, del, scale, Sm, Rm, c, kh, s);
        // softmax over j of row i = c (two lanes per row, 16 columns each)
        float m = -INFINITY;
        const int klen = key_len ? key_len[b] : L;     // opt-in padding mask: keys >= klen masked, the diagonal kept (HF)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (kh * 16 + t >= L) s[t] = -INFINITY;
            else if (kh * 16 + t >= klen && kh * 16 + t != c) s[t] = -1e30f;
            m = fmaxf(m, s[t]);
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) { s[t] = __expf(s[t] - m); sum += s[t]; }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        if (kh == 0 && c < L) lse[((long)b * n_head + h) * L + c] = m + __logf(sum);
        Frag<16> p;
        const unsigned long long mbase = ((unsigned long long)(b * n_head + h) * L + min(c, L - 1)) * L;
#pragma unroll
        for (int t = 0; t < 16; ++t) p.v[t] = s[t] * inv;      // 0 for j >= L (exp(-inf))
        if (drop.p > 0.f) {
            float msk[16];
            drop_scale_run16(drop, mbase + kh * 16, L - kh * 16, (L & 3) == 0, msk);
#pragma unroll
            for (int t = 0; t < 16; ++t) p.v[t] *= msk[t];
        }
#if T4R_ATTN_FWD_FENCE
        __builtin_amdgcn_sched_barrier(0);
#endif
        // out = P~ V : A = row layout, B = column fragment of V (k = j)
        f32x16 o = zero16();
        mfma_chain(o, p, vf);
        if (c < DH) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = xm_row(r, kh);
                if (i < L) ob[i * D + hc + c] = o[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ backward
// part: row blockIdx.x of [gridDim.x][2L*D + 2D]: d k_r (shared k_r only) | d r_w_bias | d r_r_bias,
// this head's columns (same layout the VALU kernels use; reduced by t4r_reduce_partials_launch).
template <int DH, bool SHARED_KR>
__global__ __launch_bounds__(T4R_ATTN_BWD_BOUNDS) void xlnet_attn_mfma_bwd_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ kr, const float* __restrict__ r_w_bias, const float* __restrict__ r_r_bias,
    const float* __restrict__ lse, const float* __restrict__ dout, float* __restrict__ dq,
    float* __restrict__ dk, float* __restrict__ dv, float* __restrict__ part, float* __restrict__ dkr_b, int B,
    int L, int n_head, float scale, long kr_bstride, DropCfg drop, const int* key_len) {
    __shared__ __attribute__((aligned(16))) float Sm[32 * XM_SP];
    __shared__ __attribute__((aligned(16))) float Pm[32 * XM_SP];   // dropped probabilities, parked for d v
    __shared__ __attribute__((aligned(16))) float Rm[32 * XM_RP];
    const int lane = threadIdx.x;
    const int h = blockIdx.y, hc = h * DH, D = n_head * DH;
    float acc_rw = 0.f, acc_rr = 0.f;             // column (d = c) sums of d q_ac / d q_bd over all rows and sessions
    f32x16 gkr0 = zero16(), gkr1 = zero16();      // shared k_r only: d k_r rows 0..31 / 32..63, this head's columns

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        // lane coordinates are laundered per session: everything derived from them (offsets, clamps,
        // bias values) is recomputed where it is used instead of being hoisted out of the loop and
        // kept live across all phases (that cost > 400 registers and spills)
        int c = lane & 31, kh = lane >> 5;
        asm volatile("" : "+v"(c), "+v"(kh));
        const int dc = min(c, DH - 1);
        Frag<DH / 2> rw, del;
#pragma unroll
        for (int t = 0; t < DH / 2; ++t) {
            rw.v[t] = r_w_bias[hc + kh * (DH / 2) + t];
            del.v[t] = r_r_bias[hc + kh * (DH / 2) + t] - rw.v[t];
        }
        const float rw_c = r_w_bias[hc + dc], rr_c = r_r_bias[hc + dc];
        const long tok0 = (long)b * L;
        const float* qb = q + tok0 * D;
        const float* kb = k + tok0 * D;
        const float* vb = v + tok0 * D;
        const float* gb = dout + tok0 * D;
        const float* krb = kr + (long)b * kr_bstride;
        const int ic = min(c, L - 1);
        const int roff = ic * D + hc + kh * (DH / 2);
        const bool row_ok = c < L;
        const ScoreFrags<DH> sf = load_score_frags<DH>(qb, kb, krb, D, hc, L, c, kh);
        const Frag<DH / 2> dof = row_frag<DH>(gb, roff);                 // operands of dP = dO V^T, in flight early
        const Frag<DH / 2> vf = row_frag<DH>(vb, roff);                  // lane column j = c
        const Frag<16> kcol = col_frag<16>(kb, hc + dc, D, L, kh, 0.f);  // for d q = dS K
        __builtin_amdgcn_sched_barrier(0);    // the loads are issued here; the waits sit at the uses
        float s[16];
        scores_row_layout<DH>(sf, L, rw, del, scale, Sm, Rm, c, kh, s);
        const float lrow = lse[((long)b * n_head + h) * L + ic];
        Frag<16> P;
        const int klen = key_len ? key_len[b] : L;
#pragma unroll
        for (int t = 0; t < 16; ++t)
            P.v[t] = (row_ok && kh * 16 + t < L && !(kh * 16 + t >= klen && kh * 16 + t != c)) ? __expf(s[t] - lrow) : 0.f;
        ATTN_PHASE_FENCE();   // keeps later phases' operand loads out of this one
        // (operands of the NEXT phase are requested at the start of each phase: the fences keep them here)
        Frag<16> krc[2];          // k_r column fragments for d q += d raw K_r : k = m = kh*32 + 16u + t
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < 16; ++t) krc[u].v[t] = krb[min(kh * 32 + 16 * u + t, 2 * L - 1) * D + hc + dc];
        // dP = dO V^T (accumulator layout) -> row layout through Sm
        {
            f32x16 dp = zero16();
            mfma_chain(dp, dof, vf);
#pragma unroll
            for (int r = 0; r < 16; ++r) Sm[xm_row(r, kh) * XM_SP + c] = dp[r];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        Frag<16> dS;
        {
            const unsigned long long mbase = ((unsigned long long)(b * n_head + h) * L + ic) * L;
            float drow = 0.f;
            float dpm[16], msk[16];
            if (drop.p > 0.f) {
                drop_scale_run16(drop, mbase + kh * 16, L - kh * 16, (L & 3) == 0, msk);
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) msk[t] = 1.f;
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int j = kh * 16 + t;
                dpm[t] = Sm[c * XM_SP + j] * msk[t];
                Pm[c * XM_SP + j] = P.v[t] * msk[t];
                drow += P.v[t] * dpm[t];
            }
            drow += __shfl_xor(drow, 32, 64);
#pragma unroll
            for (int t = 0; t < 16; ++t) dS.v[t] = P.v[t] * (dpm[t] - drow) * scale;
        }
        __builtin_amdgcn_wave_barrier();
        ATTN_PHASE_FENCE();   // keeps later phases' operand loads out of this one
        // d raw[i][j + L - i] = dS[i][j]  (rel_shift transposed), zero elsewhere
        for (int idx = lane; idx < 32 * XM_RP / 4; idx += 64)
            reinterpret_cast<float4*>(Rm)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (row_ok) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int j = kh * 16 + t;
                if (j < L) Rm[c * XM_RP + j + L - c] = dS.v[t];
            }
        }
        // dS and Pd by rows into Sm / (after use) for the transposed reads
#pragma unroll
        for (int t = 0; t < 16; ++t) Sm[c * XM_SP + kh * 16 + t] = dS.v[t];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        ATTN_PHASE_FENCE();   // keeps later phases' operand loads out of this one
        // d q = dS K + d raw K_r
        const Frag<16> qcol = col_frag<16>(qb, hc + dc, D, L, kh, 0.f);     // for d k, d k_r (next phases)
        f32x16 dqa = zero16(), dqb = zero16();
        {
            mfma_chain(dqa, dS, kcol);
#pragma unroll
            for (int u = 0; u < 2; ++u) {       // contraction over m (64): k = kh*32 + 16u + t
                Frag<16> dr;
#pragma unroll
                for (int t = 0; t < 16; ++t) dr.v[t] = Rm[c * XM_RP + kh * 32 + 16 * u + t];
                mfma_chain(dqb, dr, krc[u]);
            }
        }
        {
            float sa = 0.f, sb = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = xm_row(r, kh);
                sa += dqa[r]; sb += dqb[r];
                if (c < DH && i < L) dq[(tok0 + i) * D + hc + c] = dqa[r] + dqb[r];
            }
            sa += __shfl_xor(sa, 32, 64); sb += __shfl_xor(sb, 32, 64);
            acc_rw += sa; acc_rr += sb;
        }
        ATTN_PHASE_FENCE();   // keeps later phases' operand loads out of this one
        // d k = dS^T (q + r_w_bias) : A = dS^T read by columns from Sm, B = column fragment of q
        const Frag<16> gcol = col_frag<16>(gb, hc + dc, D, L, kh, 0.f);     // for d v (last phase)
        {
            Frag<16> at;
#pragma unroll
            for (int t = 0; t < 16; ++t) at.v[t] = Sm[(kh * 16 + t) * XM_SP + c];
            Frag<16> qf;
#pragma unroll
            for (int t = 0; t < 16; ++t) qf.v[t] = qcol.v[t] + rw_c;
            f32x16 g = zero16();
            mfma_chain(g, at, qf);
            if (c < DH) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = xm_row(r, kh);
                    if (j < L) dk[(tok0 + j) * D + hc + c] = g[r];
                }
            }
        }
        ATTN_PHASE_FENCE();   // keeps later phases' operand loads out of this one
        // d k_r = d raw^T (q + r_r_bias) : two 32-row tiles (m = c, c + 32)
        {
            Frag<16> qf;
#pragma unroll
            for (int t = 0; t < 16; ++t) qf.v[t] = qcol.v[t] + rr_c;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                Frag<16> at;
#pragma unroll
                for (int t = 0; t < 16; ++t) at.v[t] = Rm[(kh * 16 + t) * XM_RP + c + 32 * half];
                if (!SHARED_KR) {
                    f32x16 g = zero16();
                    mfma_chain(g, at, qf);
                    if (c < DH) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int m = xm_row(r, kh) + 32 * half;
                            if (m < 2 * L) dkr_b[((long)b * 2 * L + m) * D + hc + c] = g[r];
                        }
                    }
                } else if (half == 0) {
                    mfma_chain(gkr0, at, qf);
                } else {
                    mfma_chain(gkr1, at, qf);
                }
            }
        }
        ATTN_PHASE_FENCE();   // keeps later phases' operand loads out of this one
        // d v = P~^T dO : P~ was parked by rows in Pm, read by columns
        {
            Frag<16> at;
#pragma unroll
            for (int t = 0; t < 16; ++t) at.v[t] = Pm[(kh * 16 + t) * XM_SP + c];
            f32x16 g = zero16();
            mfma_chain(g, at, gcol);
            if (c < DH) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = xm_row(r, kh);
                    if (j < L) dv[(tok0 + j) * D + hc + c] = g[r];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    float* mypart = part + (long)blockIdx.x * (2 * L * D + 2 * D);
    const int c = lane & 31, kh = lane >> 5;
    if (c < DH) {
        if (SHARED_KR) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m0 = xm_row(r, kh);
                if (m0 < 2 * L) mypart[(long)m0 * D + hc + c] = gkr0[r];
                if (m0 + 32 < 2 * L) mypart[(long)(m0 + 32) * D + hc + c] = gkr1[r];
            }
        }
        if (kh == 0) {
            mypart[2 * L * D + hc + c] = acc_rw;
            mypart[2 * L * D + D + hc + c] = acc_rr;
        }
    }
}

int t4r_reduce_partials_launch(hipStream_t st, const float* part, int nblocks, float* o0, int n0, int a0,
                               float* o1, int n1, int a1, float* o2, int n2, int a2);

// internal entry points used by xlnet_attn.hip's dispatch (same argument meaning as the VALU kernels)
int t4r_xlnet_attn_mfma_ok(int L, int d_head) { return L >= 1 && L <= 32 && (d_head == 16 || d_head == 32); }
int t4r_xlnet_attn_mfma_blocks(int B) { return B < 1024 ? B : 1024; }

int t4r_xlnet_attn_mfma_fwd(hipStream_t st, const float* q, const float* k, const float* v, const float* kr,
                            const float* rw, const float* rr, float* out, float* lse, int B, int L, int n_head,
                            int d_head, float scale, long kr_bstride, DropCfg drop, const int* key_len) {
    const dim3 grid(B < 4096 ? B : 4096, n_head), block(64);
    if (d_head == 32)
        hipLaunchKernelGGL(xlnet_attn_mfma_fwd_kernel<32>, grid, block, 0, st, q, k, v, kr, rw, rr, out, lse, B, L,
                           n_head, scale, kr_bstride, drop, key_len);
    else
        hipLaunchKernelGGL(xlnet_attn_mfma_fwd_kernel<16>, grid, block, 0, st, q, k, v, kr, rw, rr, out, lse, B, L,
                           n_head, scale, kr_bstride, drop, key_len);
    T4R_LAUNCH_CHECK();
    return 0;
}

int t4r_xlnet_attn_mfma_bwd(hipStream_t st, const float* q, const float* k, const float* v, const float* kr,
                            const float* rw, const float* rr, const float* lse, const float* dout, float* dq,
                            float* dk, float* dv, float* part, float* dkr, float* d_rw, float* d_rr, int B, int L,
                            int n_head, int d_head, float scale, long kr_bstride, DropCfg drop, const int* key_len) {
    const int D = n_head * d_head;
    // (one (session, head) unit per wave and 1024 x n_head waves: at four heads exactly one residency of 256 CUs.  Shrinking the
    //  grid to the CU budget was tried (round 5) and is worse: the unit is a whole session, so 1024 sessions on 960 workgroups
    //  are two rounds for 64 of them anyway -- the occupier curve went 1.25x -> 1.5x.)
    const int gx = t4r_xlnet_attn_mfma_blocks(B);
    const dim3 grid(gx, n_head), block(64);
    float* dkr_b = kr_bstride > 0 ? dkr : nullptr;
#define T4R_BWD(DHV, SH)                                                                                      \
    hipLaunchKernelGGL((xlnet_attn_mfma_bwd_kernel<DHV, SH>), grid, block, 0, st, q, k, v, kr, rw, rr, lse, dout, dq, \
                       dk, dv, part, dkr_b, B, L, n_head, scale, kr_bstride, drop, key_len)
    if (d_head == 32) { if (dkr_b) T4R_BWD(32, false); else T4R_BWD(32, true); }
    else { if (dkr_b) T4R_BWD(16, false); else T4R_BWD(16, true); }
#undef T4R_BWD
    T4R_LAUNCH_CHECK();
    return t4r_reduce_partials_launch(st, part, gx, kr_bstride > 0 ? nullptr : dkr, 2 * L * D, 0, d_rw, D, 1, d_rr,
                                      D, 1);
}
