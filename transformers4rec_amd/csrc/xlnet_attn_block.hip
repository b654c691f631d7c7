// The ATTENTION HALF of an XLNet layer as ONE kernel per direction (round 4), on the exact-fp32 matrix instruction
// v_mfma_f32_16x16x4_f32 (a k-ordered fmaf chain, bit for bit: the reference's own fp32 arithmetic):
//
//   forward   h tile -> q | k | v projections -> relative attention core (ac + shifted bd, softmax, Philox dropout, P v)
//             -> o-projection + dropout + residual + LayerNorm -> h1
//             HF modeling_xlnet.py XLNetRelativeAttention.forward :245-282 (g = None branch: :251-258 q/k/v heads,
//             :266 k_head_r [precomputed: k_r], rel_attn_core :96-140, rel_shift_bnij :86-94), post_attention :142-152,
//             as called by transformers4rec/torch/block/transformer.py:179-199
//   backward  (second half of this file) LayerNorm backward -> d attn_out -> d attn_vec -> attention core backward
//             (d q, d k, d v, d k_r, d r_w_bias, d r_r_bias) -> d h
//
// One workgroup owns S whole sessions (S L <= 80 token rows: 4 sessions of L = 20 = 256 workgroups = one per CU at
// batch 1024) and D / 16 waves.  Every product is "transposed" as in xlnet_fused.h: the A operand is the matrix whose
// rows are the product's OUTPUT FEATURES (a weight row, a key row, a value column), the B operand holds the tokens, so
// an accumulator lane owns 4 consecutive features of ONE token and every store is 16 bytes.  k-slots are permuted --
// step (c, e) of a contraction takes k = 16 c + 4 g + e from lane group g = lane >> 4 on BOTH operands -- so an operand
// whose k runs along a row of a row-major matrix is one 16-byte read per four matrix instructions, and the transposed
// score tile S^T[j][i] in accumulator layout (lane (i = lane & 15, g) holds keys j = 16 jt + 4 g + r) IS the B operand of
// the contraction over the keys that follows (P v, d q, ...): the softmax never leaves the registers.
//
//   phase P   wave w: features 16 w .. 16 w + 15 of q, k and v for all token rows of the tile (A = three weight-row
//             fragments held in registers, B = the tile's h rows, staged once in the v columns of the LDS tile)
//             -> LDS tile [row][q | k | v] and the saved qkv
//   phase A   one (session, head) unit per wave at a time, operands from the LDS tile, k_r rows from memory; the
//             relative shift is a gather from a 16 x 64 exchange buffer per wave; attn_vec replaces the unit's own q
//   phase O   wave w: features 16 w .. of attn_vec @ o^T (+ dropout + residual + LayerNorm over the row, the row sums
//             exchanged between the waves through LDS as in xlnet_oproj_ln_kernel)
//
// Why fp32 matrix instructions here while the feed-forward block and the head run on the two-way fp16 split: the
// attention half is small (12 GFLOP forward per step at BASELINE configs[1]) and was bound by launch boundaries, operand
// cutting and HBM round trips of q / k / v, not by the matrix pipe -- four launches (projection, core, o-projection +
// LayerNorm, with q, k, v, attn_vec making a round trip through HBM between them) took 55 us per layer for 22 us of
// fp32 matrix time.  In one kernel the matrix pipe is the bound: measured numbers in DESIGN.md (round 4).
#include "xlnet_fused.h"

namespace {

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float4 lds4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void put4(float (&d)[4], float4 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
__device__ __forceinline__ float4 f4(const f32x4& a) { return make_float4(a[0], a[1], a[2], a[3]); }
// Stores of what only the BACKWARD pass reads (q | k | v, attn_vec, the o-projection before dropout): written through to
// memory (sc1) instead of left dirty in the XCD's L2.  A kernel boundary writes every dirty L2 line back (the eight L2s
// are not coherent with each other), so 50 MB left dirty here would come due at once in front of the next kernel;
// written through they leave the chip while the matrix pipe works.  T4R_AB_STORE: 0 plain, 1 sc1 (default), 2 nt.
#ifndef T4R_AB_STORE
#define T4R_AB_STORE 1
#endif
__device__ __forceinline__ void st4_stream(float* p, const f32x4& v) {
#if T4R_AB_STORE == 1
    // (s_nop: the wait states between a 128-bit store and a later write of its data registers -- hipcc pads nothing inside asm)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#elif T4R_AB_STORE == 2
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
#else
    *reinterpret_cast<f32x4*>(p) = v;
#endif
}
// this wave's LDS traffic is complete and visible to its own lanes (wave-private buffers: no workgroup barrier)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
}

// Phase stamps (tools/attn_block_bench.py, a variant build with -DT4R_AB_STAMPS; never in the product library): lane 0
// of every wave leaves s_memtime at the phase borders in a debug buffer
#ifdef T4R_AB_STAMPS
static thread_local long long* g_ab_stamps = nullptr;
extern "C" void t4r_debug_ab_stamps(void* buf) { g_ab_stamps = (long long*)buf; }
#define AB_STAMP_(k) do { if (p.stamps && lane == 0) p.stamps[((long)blockIdx.x * (blockDim.x >> 6) + w) * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#if T4R_AB_STAMPS == 2      /* detail of phase P: after every token block */
#define AB_STAMP(k) do { if ((k) == 0) AB_STAMP_(0); } while (0)
#define AB_STAMP_P(k) AB_STAMP_(k)
#else
#define AB_STAMP(k) AB_STAMP_(k)
#define AB_STAMP_P(k)
#endif
#if T4R_AB_STAMPS == 3      /* detail of the backward's phase 3: first unit, first query block */
#define AB_STAMP_3(k, cond) do { if (cond) AB_STAMP_(k); } while (0)
#define AB_STAMP_B(k)
#else
#define AB_STAMP_3(k, cond)
#define AB_STAMP_B(k) AB_STAMP_(k)
#endif
#else
#define AB_STAMP_3(k, cond)
#define AB_STAMP_B(k)
#define AB_STAMP(k)
#define AB_STAMP_P(k)
#define AB_STAMP_(k)
#endif

#ifndef T4R_CORE_NB
#define T4R_CORE_NB 8
#endif
#ifndef T4R_CORE_SLOTS
#define T4R_CORE_SLOTS 512
#endif
// workgroup barrier that orders LDS traffic only (__syncthreads also waits for the wave's stores to memory)
__device__ __forceinline__ void lds_barrier() {
#ifdef T4R_CORE_FULLBAR
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}
constexpr int AB_RT = 80;       // token rows of a workgroup tile (5 blocks of 16)
constexpr int AB_R = AB_RT / 16;
constexpr int AB_PR = 68;       // pitch of the per-wave 16 x 64 exchange buffer (PR - 1 odd: the shifted gather spreads over the banks)

}  // namespace

struct AttnBlockFwd {
    const float* h;          // [T, D] layer input
    const float* wqkvT;      // [3 D][D]  W_z^T (carve_planes_32)
    const float* wo;         // [D][D]    o as stored: rows = output feature, columns = (head, d)
    const float* kr;         // [2 L, D] shared or [B, 2 L, D] per session (kr_bstride = 2 L D)
    long kr_bstride;
    const float *rw, *rr;    // r_w_bias, r_r_bias [D]
    const float *gamma, *beta;
    float *qkv, *av, *lse;   // saved for the backward: [3][T][D], [T][D], [B][n_head][L]
    float *ao, *mean, *rstd; // o-projection output before dropout [T][D], LayerNorm statistics [T] (all NULL: inference)
    float* h1;               // [T, D]
    const int* key_len;      // optional [B]
    int B, L, S;
    long T;
    float scale, eps;
    DropCfg drop_p, drop_o;  // attention probabilities, attention output
#ifdef T4R_AB_STAMPS
    long long* stamps;
#endif
};

template <int V>
struct IC { static constexpr int value = V; };

template <int D, int DH>
__global__ __launch_bounds__(D * 4) void xlnet_attn_block_fwd_kernel(AttnBlockFwd p) {
    constexpr int NW = D / 16, NH = D / DH, KC = D / 16, HC = DH / 16, PQ = 3 * D + 4;
    extern __shared__ float smem[];
    float* tile = smem;                               // [AB_RT][PQ]: q | k | v of the tile's rows (q becomes attn_vec)
    float* xbuf = smem + AB_RT * PQ;                  // [NW][16][AB_PR] exchange buffers; later the LayerNorm row sums
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, g = lane >> 4;
    const int L = p.L;
    const int b0 = blockIdx.x * p.S, nb = min(p.S, p.B - b0), rows = nb * L;
    const long t0 = (long)b0 * L, TD = p.T * D;

    // ------------------------------------------------------------------------------------------------ phase P
    // All five token blocks are always computed (rows past the tile's sessions are clamped copies: finite, never stored):
    // a guard per block puts a scalar branch between every pair of matrix instructions (measured: 40 instead of 32 cycles
    // per instruction).
    AB_STAMP(0);
    {
        // The tile's h rows enter the chip ONCE, coalesced, into the v columns of the LDS tile (free until the v products
        // are written): every wave needs all of them, and read per wave from memory they were 8 x 40 KB per workgroup.
        // Order of the requests = order of use: the h rows, then the q and k weight fragments k-chunk by k-chunk (the
        // products walk the chunks in the same order and start on the first fragments to arrive: the workgroup's 236 KB
        // cross a load path that delivers ~15 B per cycle and CU while every CU opens at once), the v weights last.
        constexpr int NST = (AB_RT * (D / 4) + NW * 64 - 1) / (NW * 64);
        float4 hstage[NST];
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int idx = min(tid + i * NW * 64, AB_RT * (D / 4) - 1);
            const int row = idx / (D / 4), c4 = (idx - row * (D / 4)) * 4;
            hstage[i] = ld4(p.h + min(t0 + row, p.T - 1) * D + c4);
        }
        float a[3][4 * KC];
        auto load_a = [&](int z, int c) __attribute__((always_inline)) {
            float t4[4];
            put4(t4, ld4(p.wqkvT + (long)(z * D + 16 * w + n) * D + 16 * c + 4 * g));
#pragma unroll
            for (int e = 0; e < 4; ++e) a[z][4 * c + e] = t4[e];
        };
#pragma unroll
        for (int c = 0; c < KC; ++c) { load_a(0, c); load_a(1, c); }
#pragma unroll
        for (int c = 0; c < KC; ++c) load_a(2, c);
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int idx = tid + i * NW * 64;
            const int row = idx / (D / 4), c4 = (idx - row * (D / 4)) * 4;
            if (idx < AB_RT * (D / 4)) *reinterpret_cast<float4*>(tile + row * PQ + 2 * D + c4) = hstage[i];
        }
        __syncthreads();
        AB_STAMP_P(1);
        const float* hb = tile + n * PQ + 2 * D + 4 * g;
        auto load_x = [&](int c, float (&x)[AB_R][4]) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < AB_R; ++r) put4(x[r], lds4(hb + 16 * r * PQ + 16 * c));
        };
        // pass 1: q and k together -- ten independent accumulator chains (two matrices x five token blocks), k-chunk outer;
        // the LDS rows of the next chunk are requested before this chunk's products
        {
            f32x4 acc[2][AB_R];
#pragma unroll
            for (int z = 0; z < 2; ++z)
#pragma unroll
                for (int r = 0; r < AB_R; ++r) acc[z][r] = zero4();
            float xb[2][AB_R][4];
            load_x(0, xb[0]);
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                if (c + 1 < KC) load_x(c + 1, xb[(c + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < AB_R; ++r) {
                        acc[0][r] = mfma4(a[0][4 * c + e], xb[c & 1][r][e], acc[0][r]);
                        acc[1][r] = mfma4(a[1][4 * c + e], xb[c & 1][r][e], acc[1][r]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int z = 0; z < 2; ++z)
#pragma unroll
                for (int r = 0; r < AB_R; ++r) {
                    const int tok = 16 * r + n;
                    *reinterpret_cast<float4*>(tile + tok * PQ + z * D + 16 * w + 4 * g) = f4(acc[z][r]);
                    if (tok < rows) st4_stream(p.qkv + z * TD + (t0 + tok) * D + 16 * w + 4 * g, acc[z][r]);
                }
        }
        AB_STAMP_P(2);
        // pass 2: v, kept in its accumulators until every wave has read the h rows it replaces
        f32x4 accv[AB_R];
#pragma unroll
        for (int r = 0; r < AB_R; ++r) accv[r] = zero4();
        {
            float xb[2][AB_R][4];
            load_x(0, xb[0]);
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                if (c + 1 < KC) load_x(c + 1, xb[(c + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < AB_R; ++r) accv[r] = mfma4(a[2][4 * c + e], xb[c & 1][r][e], accv[r]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        AB_STAMP_P(3);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < AB_R; ++r) {
            const int tok = 16 * r + n;
            *reinterpret_cast<float4*>(tile + tok * PQ + 2 * D + 16 * w + 4 * g) = f4(accv[r]);
            if (tok < rows) st4_stream(p.qkv + 2 * TD + (t0 + tok) * D + 16 * w + 4 * g, accv[r]);
        }
        AB_STAMP_P(4);
        AB_STAMP_P(5);
    }
    AB_STAMP(1);
    __syncthreads();
    AB_STAMP(2);

    // the o weight rows of phase O are requested here: they arrive under the attention phase
    float wof[4 * KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        float t4[4];
        put4(t4, ld4(p.wo + (long)(16 * w + n) * D + 16 * c + 4 * g));
#pragma unroll
        for (int e = 0; e < 4; ++e) wof[4 * c + e] = t4[e];
    }
    // ------------------------------------------------------------------------------------------------ phase A
    // MT = blocks of 16 relative positions (ceil(2 L / 16)), JT = (MT + 1) / 2 = blocks of 16 keys / queries: compile-time
    // in the body (a run-time guard per block is a branch between the matrix instructions).
    auto phase_a = [&](auto MTc) __attribute__((always_inline)) {
        constexpr int MT = decltype(MTc)::value, JT = (MT + 1) / 2;
        float* Rm = xbuf + w * 16 * AB_PR;
        const bool aligned = (L & 3) == 0;
        for (int u = w; u < nb * NH; u += NW) {
            const int s = u / NH, hh = u - s * NH, b = b0 + s, r0 = s * L, hc = hh * DH;
            const float* krb = p.kr + (long)b * p.kr_bstride;
            const int klen = p.key_len ? p.key_len[b] : L;
            // The A operands of the unit do not depend on the query block: loaded once, the k_r rows (memory) first.  All
            // loads are unconditional with clamped rows (a guarded load forces waits on everything in flight).
            float krf[MT][4 * HC], kf[JT][4 * HC], vf[JT][4][HC];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int c = 0; c < HC; ++c) {
                    float t4[4];
                    put4(t4, ld4(krb + (long)min(16 * mt + n, 2 * L - 1) * D + hc + 16 * c + 4 * g));
#pragma unroll
                    for (int e = 0; e < 4; ++e) krf[mt][4 * c + e] = t4[e];
                }
            float rwv[4 * HC], rrv[4 * HC];
#pragma unroll
            for (int c = 0; c < HC; ++c) {
                float t4[4];
                put4(t4, ld4(p.rw + hc + 16 * c + 4 * g));
#pragma unroll
                for (int e = 0; e < 4; ++e) rwv[4 * c + e] = t4[e];
                put4(t4, ld4(p.rr + hc + 16 * c + 4 * g));
#pragma unroll
                for (int e = 0; e < 4; ++e) rrv[4 * c + e] = t4[e];
            }
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int c = 0; c < HC; ++c) {
                    float t4[4];
                    put4(t4, lds4(tile + (r0 + min(16 * jt + n, L - 1)) * PQ + D + hc + 16 * c + 4 * g));
#pragma unroll
                    for (int e = 0; e < 4; ++e) kf[jt][4 * c + e] = t4[e];
                }
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int dt = 0; dt < HC; ++dt)
                        vf[jt][e][dt] = tile[(r0 + min(16 * jt + 4 * g + e, L - 1)) * PQ + 2 * D + hc + 16 * dt + n];
            // scores of one query block: S^T = k (q + r_w_bias)^T into sT, raw^T = k_r (q + r_r_bias)^T into rT
            auto scores = [&](int it, f32x4 (&sT)[JT], f32x4 (&rT)[MT]) __attribute__((always_inline)) {
                const int ic = min(16 * it + n, L - 1);
                float bw[4 * HC], br[4 * HC];
#pragma unroll
                for (int c = 0; c < HC; ++c) {
                    float t4[4];
                    put4(t4, lds4(tile + (r0 + ic) * PQ + hc + 16 * c + 4 * g));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bw[4 * c + e] = t4[e] + rwv[4 * c + e]; br[4 * c + e] = t4[e] + rrv[4 * c + e]; }
                }
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) sT[jt] = zero4();
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) rT[mt] = zero4();
#pragma unroll
                for (int i4 = 0; i4 < 4 * HC; ++i4) {
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) sT[jt] = mfma4(kf[jt][i4], bw[i4], sT[jt]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) rT[mt] = mfma4(krf[mt][i4], br[i4], rT[mt]);
                }
            };
            f32x4 sT[2][JT], rT[2][MT];
            scores(0, sT[0], rT[0]);
#pragma unroll
            for (int it = 0; it < JT; ++it) {
                const int i = 16 * it + n, ic = min(i, L - 1);
                // raw^T of this block -> exchange buffer -> the shifted gather bd[i][j] = raw[i][j + L - i]
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<float4*>(Rm + n * AB_PR + 16 * mt + 4 * g) = f4(rT[it][mt]);
                wave_lds_sync();
                float pv[JT][4];
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int jj = min(16 * jt + 4 * g + r, L - 1);
                        pv[jt][r] = sT[it][jt][r] + Rm[n * AB_PR + jj + L - ic];
                    }
                wave_lds_sync();                      // the exchange buffer is rewritten by the next query block
                // the NEXT query block's score products are issued here: they run under this block's softmax
                if (it + 1 < JT) scores(it + 1, sT[(it + 1) & 1], rT[(it + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                float mx = -INFINITY;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = 16 * jt + 4 * g + r;
                        float sv = pv[jt][r] * p.scale;
                        if (j >= L) sv = -INFINITY;
                        else if (j >= klen && j != i) sv = -1e30f;                            // opt-in padding mask, diagonal kept (HF)
                        pv[jt][r] = sv;
                        mx = fmaxf(mx, sv);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                float sum = 0.f;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { pv[jt][r] = __expf(pv[jt][r] - mx); sum += pv[jt][r]; }
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                const float inv = 1.f / sum;
                if (g == 0 && i < L) p.lse[((long)b * NH + hh) * L + i] = mx + __logf(sum);
                const unsigned long long mbase = ((unsigned long long)(b * NH + hh) * L + ic) * L;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    const int j0 = 16 * jt + 4 * g;
                    float m[4] = {1.f, 1.f, 1.f, 1.f};
                    if (p.drop_p.p > 0.f && j0 < L) {
                        if (aligned) {
                            const float4 f = drop_scale4(p.drop_p, mbase + j0);
                            m[0] = f.x; m[1] = f.y; m[2] = f.z; m[3] = f.w;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (j0 + r < L) m[r] = drop_scale(p.drop_p, mbase + j0 + r);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[jt][r] = pv[jt][r] * inv * m[r];          // 0 for j >= L (exp(-inf))
                }
                // attn_vec^T[d][i] = sum_j v[j][d] P~[i][j]:  A = a value column (k = j), B = the probabilities in place
                f32x4 o[HC];
#pragma unroll
                for (int dt = 0; dt < HC; ++dt) o[dt] = zero4();
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int dt = 0; dt < HC; ++dt) o[dt] = mfma4(vf[jt][e][dt], pv[jt][e], o[dt]);
                if (i < L) {
#pragma unroll
                    for (int dt = 0; dt < HC; ++dt) {
                        *reinterpret_cast<float4*>(tile + (r0 + i) * PQ + hc + 16 * dt + 4 * g) = f4(o[dt]);   // over the unit's own q
                        st4_stream(p.av + (t0 + r0 + i) * D + hc + 16 * dt + 4 * g, o[dt]);
                    }
                }
            }
        }
    };
    switch ((2 * L + 15) / 16) {
        case 1: phase_a(IC<1>()); break;
        case 2: phase_a(IC<2>()); break;
        case 3: phase_a(IC<3>()); break;
        default: phase_a(IC<4>()); break;
    }
    AB_STAMP(3);
    __syncthreads();
    AB_STAMP(4);

    // ------------------------------------------------------------------------------------------------ phase O
    {
        const bool train = p.ao != nullptr;
        // the residual rows of the epilogue are requested before the products (their latency would sit in front of the LayerNorm)
        float4 hres[AB_R];
#pragma unroll
        for (int r = 0; r < AB_R; ++r) hres[r] = ld4(p.h + min(t0 + 16 * r + n, p.T - 1) * D + 16 * w + 4 * g);
        f32x4 acc[AB_R];
#pragma unroll
        for (int r = 0; r < AB_R; ++r) acc[r] = zero4();
        {
            const float* vb = tile + n * PQ + 4 * g;
            float xb[2][AB_R][4];
#pragma unroll
            for (int r = 0; r < AB_R; ++r) put4(xb[0][r], lds4(vb + 16 * r * PQ));
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                if (c + 1 < KC) {
#pragma unroll
                    for (int r = 0; r < AB_R; ++r) put4(xb[(c + 1) & 1][r], lds4(vb + 16 * r * PQ + 16 * (c + 1)));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < AB_R; ++r) acc[r] = mfma4(wof[4 * c + e], xb[c & 1][r][e], acc[r]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        AB_STAMP(5);
        // epilogue (xlnet_oproj_ln_kernel's): dropout on the projection, residual, LayerNorm over the row; a row's D
        // features live in NW waves x 4 lane groups
        float* sh_red = xbuf;                         // [2][NW][AB_RT]
        const int f0 = 16 * w + 4 * g;
        const float4 gam = ld4(p.gamma + f0), bet = ld4(p.beta + f0);
        float4 x[AB_R];
        float sum[AB_R];
#pragma unroll
        for (int r = 0; r < AB_R; ++r) {
            const int tok = 16 * r + n;
            const long t = t0 + tok, tc = min(t, p.T - 1);
            float4 v = f4(acc[r]);
            const bool live = tok < rows;
            if (train && live) st4_stream(p.ao + t * D + f0, acc[r]);
            if (p.drop_o.p > 0.f) {
                const float4 m = drop_scale4(p.drop_o, (unsigned long long)tc * D + f0);
                v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
            }
            v.x += hres[r].x; v.y += hres[r].y; v.z += hres[r].z; v.w += hres[r].w;
            x[r] = v;
            float sm = (v.x + v.y) + (v.z + v.w);
            sm += __shfl_xor(sm, 16, 64);
            sm += __shfl_xor(sm, 32, 64);
            sum[r] = sm;
        }
        if (g == 0) {
#pragma unroll
            for (int r = 0; r < AB_R; ++r) sh_red[w * AB_RT + 16 * r + n] = sum[r];
        }
        __syncthreads();
        float mu[AB_R];
#pragma unroll
        for (int r = 0; r < AB_R; ++r) {
            float sm = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) sm += sh_red[ww * AB_RT + 16 * r + n];
            mu[r] = sm * (1.0f / D);
            const float dx = x[r].x - mu[r], dy = x[r].y - mu[r], dz = x[r].z - mu[r], dw = x[r].w - mu[r];
            float q = (dx * dx + dy * dy) + (dz * dz + dw * dw);
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            sum[r] = q;
        }
        float* sh_red2 = sh_red + NW * AB_RT;
        if (g == 0) {
#pragma unroll
            for (int r = 0; r < AB_R; ++r) sh_red2[w * AB_RT + 16 * r + n] = sum[r];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < AB_R; ++r) {
            const int tok = 16 * r + n;
            const long t = t0 + tok;
            float q = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) q += sh_red2[ww * AB_RT + 16 * r + n];
            const float rs = rsqrtf(q * (1.0f / D) + p.eps);
            if (tok < rows) {
                st4(p.h1 + t * D + f0, make_float4((x[r].x - mu[r]) * rs * gam.x + bet.x, (x[r].y - mu[r]) * rs * gam.y + bet.y,
                                                   (x[r].z - mu[r]) * rs * gam.z + bet.z, (x[r].w - mu[r]) * rs * gam.w + bet.w));
                if (train && w == 0 && g == 0) { p.mean[t] = mu[r]; p.rstd[t] = rs; }
            }
        }
    }
    AB_STAMP(6);
}

// ------------------------------------------------------------------------------------------------ forward, two workgroups per CU
// The same kernel laid out for TWO co-resident workgroups per CU (d_model 128 only; T4R_XLNET_ATTN_BLOCK2=1): a workgroup owns
// 40 token rows (2 sessions of L = 20; three 16-row blocks, the last one half padding) and has 4 waves, each owning TWO blocks of
// 16 features.  LDS 62 KB tile + 17 KB exchange buffers = 79.5 KB: two fit the CU's 160 KB, so the phases of one workgroup --
// which run in series and leave the matrix pipe idle during loads, softmax and the LayerNorm epilogue -- overlap the other's.
// Arithmetic and saved tensors identical to the kernel above (same contraction order per output element).
// MEASURED (round 4, B 1024 / L 20 / 4 heads): 60.9 us against 54.7 us with dropout, 55.4 against 51.4 without -- NOT the default.
// 213 VGPRs, no spills, both workgroups resident; but they are dispatched together and walk the same phases at the same
// time (they contend for the matrix pipe in phase P and idle together in the epilogues), the 40 -> 48 row padding adds 20 % of
// matrix work and every CU pulls the weights twice.  Overlap needs workgroups that are OUT of phase, not merely two of them.
constexpr int AB2_RT = 40, AB2_R = 3, AB2_FB = 2;

template <int DH>
__global__ __launch_bounds__(256, 2) void xlnet_attn_block2_fwd_kernel(AttnBlockFwd p) {
    constexpr int D = 128, FB = AB2_FB, NW = D / (16 * FB), NH = D / DH, KC = D / 16, HC = DH / 16, PQ = 3 * D + 4, RT = AB2_RT, R = AB2_R;
    extern __shared__ float smem[];
    float* tile = smem;                               // [RT][PQ]
    float* xbuf = smem + RT * PQ;                     // [NW][16][AB_PR]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, g = lane >> 4;
    const int L = p.L;
    const int b0 = blockIdx.x * p.S, nb = min(p.S, p.B - b0), rows = nb * L;
    const long t0 = (long)b0 * L, TD = p.T * D;
    // LDS row of token block r, lane n (rows past the tile are clamped copies: finite, never stored)
    int rowc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) rowc[r] = min(16 * r + n, RT - 1);

    // ------------------------------------------------------------------------------------------------ phase P
    {
        constexpr int NST = (RT * (D / 4) + NW * 64 - 1) / (NW * 64);
        float4 hstage[NST];
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int idx = min(tid + i * NW * 64, RT * (D / 4) - 1);
            const int row = idx / (D / 4), c4 = (idx - row * (D / 4)) * 4;
            hstage[i] = ld4(p.h + min(t0 + row, p.T - 1) * D + c4);
        }
        float a[FB][2][4 * KC];
        auto load_a = [&](float (&dst)[4 * KC], int z, int fb) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                float t4[4];
                put4(t4, ld4(p.wqkvT + (long)(z * D + 16 * (FB * w + fb) + n) * D + 16 * c + 4 * g));
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[4 * c + e] = t4[e];
            }
        };
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) { load_a(a[fb][0], 0, fb); load_a(a[fb][1], 1, fb); }
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int idx = tid + i * NW * 64;
            const int row = idx / (D / 4), c4 = (idx - row * (D / 4)) * 4;
            if (idx < RT * (D / 4)) *reinterpret_cast<float4*>(tile + row * PQ + 2 * D + c4) = hstage[i];
        }
        __syncthreads();
        auto load_x = [&](int c, float (&x)[R][4]) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < R; ++r) put4(x[r], lds4(tile + rowc[r] * PQ + 2 * D + 4 * g + 16 * c));
        };
        // q and k of this wave's two feature blocks, one block at a time: six independent accumulator chains each
        float av_[FB][4 * KC];
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) {
            f32x4 acc[2][R];
#pragma unroll
            for (int z = 0; z < 2; ++z)
#pragma unroll
                for (int r = 0; r < R; ++r) acc[z][r] = zero4();
            float xb[2][R][4];
            load_x(0, xb[0]);
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                if (c + 1 < KC) load_x(c + 1, xb[(c + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        acc[0][r] = mfma4(a[fb][0][4 * c + e], xb[c & 1][r][e], acc[0][r]);
                        acc[1][r] = mfma4(a[fb][1][4 * c + e], xb[c & 1][r][e], acc[1][r]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            // the v weights of this block are requested here: they arrive under the other block's products / the stores
            load_a(av_[fb], 2, fb);
            const int f0 = 16 * (FB * w + fb) + 4 * g;
#pragma unroll
            for (int z = 0; z < 2; ++z)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int tok = 16 * r + n;
                    if (tok < RT) *reinterpret_cast<float4*>(tile + tok * PQ + z * D + f0) = f4(acc[z][r]);
                    if (tok < rows) st4_stream(p.qkv + z * TD + (t0 + tok) * D + f0, acc[z][r]);
                }
        }
        // v, kept in its accumulators until every wave has read the h rows it replaces
        f32x4 accv[FB][R];
#pragma unroll
        for (int fb = 0; fb < FB; ++fb)
#pragma unroll
            for (int r = 0; r < R; ++r) accv[fb][r] = zero4();
        {
            float xb[2][R][4];
            load_x(0, xb[0]);
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                if (c + 1 < KC) load_x(c + 1, xb[(c + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        accv[0][r] = mfma4(av_[0][4 * c + e], xb[c & 1][r][e], accv[0][r]);
                        accv[1][r] = mfma4(av_[1][4 * c + e], xb[c & 1][r][e], accv[1][r]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int fb = 0; fb < FB; ++fb)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int tok = 16 * r + n, f0 = 16 * (FB * w + fb) + 4 * g;
                if (tok < RT) *reinterpret_cast<float4*>(tile + tok * PQ + 2 * D + f0) = f4(accv[fb][r]);
                if (tok < rows) st4_stream(p.qkv + 2 * TD + (t0 + tok) * D + f0, accv[fb][r]);
            }
    }
    __syncthreads();

    // ------------------------------------------------------------------------------------------------ phase A (as above, NW = 4)
    auto phase_a = [&](auto MTc) __attribute__((always_inline)) {
        constexpr int MT = decltype(MTc)::value, JT = (MT + 1) / 2;
        float* Rm = xbuf + w * 16 * AB_PR;
        const bool aligned = (L & 3) == 0;
        for (int u = w; u < nb * NH; u += NW) {
            const int s = u / NH, hh = u - s * NH, b = b0 + s, r0 = s * L, hc = hh * DH;
            const float* krb = p.kr + (long)b * p.kr_bstride;
            const int klen = p.key_len ? p.key_len[b] : L;
            float krf[MT][4 * HC], kf[JT][4 * HC], vf[JT][4][HC];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int c = 0; c < HC; ++c) {
                    float t4[4];
                    put4(t4, ld4(krb + (long)min(16 * mt + n, 2 * L - 1) * D + hc + 16 * c + 4 * g));
#pragma unroll
                    for (int e = 0; e < 4; ++e) krf[mt][4 * c + e] = t4[e];
                }
            float rwv[4 * HC], rrv[4 * HC];
#pragma unroll
            for (int c = 0; c < HC; ++c) {
                float t4[4];
                put4(t4, ld4(p.rw + hc + 16 * c + 4 * g));
#pragma unroll
                for (int e = 0; e < 4; ++e) rwv[4 * c + e] = t4[e];
                put4(t4, ld4(p.rr + hc + 16 * c + 4 * g));
#pragma unroll
                for (int e = 0; e < 4; ++e) rrv[4 * c + e] = t4[e];
            }
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int c = 0; c < HC; ++c) {
                    float t4[4];
                    put4(t4, lds4(tile + (r0 + min(16 * jt + n, L - 1)) * PQ + D + hc + 16 * c + 4 * g));
#pragma unroll
                    for (int e = 0; e < 4; ++e) kf[jt][4 * c + e] = t4[e];
                }
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int dt = 0; dt < HC; ++dt)
                        vf[jt][e][dt] = tile[(r0 + min(16 * jt + 4 * g + e, L - 1)) * PQ + 2 * D + hc + 16 * dt + n];
            auto scores = [&](int it, f32x4 (&sT)[JT], f32x4 (&rT)[MT]) __attribute__((always_inline)) {
                const int ic = min(16 * it + n, L - 1);
                float bw[4 * HC], br[4 * HC];
#pragma unroll
                for (int c = 0; c < HC; ++c) {
                    float t4[4];
                    put4(t4, lds4(tile + (r0 + ic) * PQ + hc + 16 * c + 4 * g));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bw[4 * c + e] = t4[e] + rwv[4 * c + e]; br[4 * c + e] = t4[e] + rrv[4 * c + e]; }
                }
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) sT[jt] = zero4();
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) rT[mt] = zero4();
#pragma unroll
                for (int i4 = 0; i4 < 4 * HC; ++i4) {
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) sT[jt] = mfma4(kf[jt][i4], bw[i4], sT[jt]);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) rT[mt] = mfma4(krf[mt][i4], br[i4], rT[mt]);
                }
            };
            f32x4 sT[2][JT], rT[2][MT];
            scores(0, sT[0], rT[0]);
#pragma unroll
            for (int it = 0; it < JT; ++it) {
                const int i = 16 * it + n, ic = min(i, L - 1);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<float4*>(Rm + n * AB_PR + 16 * mt + 4 * g) = f4(rT[it][mt]);
                wave_lds_sync();
                float pv[JT][4];
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int jj = min(16 * jt + 4 * g + r, L - 1);
                        pv[jt][r] = sT[it][jt][r] + Rm[n * AB_PR + jj + L - ic];
                    }
                wave_lds_sync();
                if (it + 1 < JT) scores(it + 1, sT[(it + 1) & 1], rT[(it + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                float mx = -INFINITY;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = 16 * jt + 4 * g + r;
                        float sv = pv[jt][r] * p.scale;
                        if (j >= L) sv = -INFINITY;
                        else if (j >= klen && j != i) sv = -1e30f;
                        pv[jt][r] = sv;
                        mx = fmaxf(mx, sv);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                float sum = 0.f;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { pv[jt][r] = __expf(pv[jt][r] - mx); sum += pv[jt][r]; }
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                const float inv = 1.f / sum;
                if (g == 0 && i < L) p.lse[((long)b * NH + hh) * L + i] = mx + __logf(sum);
                const unsigned long long mbase = ((unsigned long long)(b * NH + hh) * L + ic) * L;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    const int j0 = 16 * jt + 4 * g;
                    float m[4] = {1.f, 1.f, 1.f, 1.f};
                    if (p.drop_p.p > 0.f && j0 < L) {
                        if (aligned) {
                            const float4 f = drop_scale4(p.drop_p, mbase + j0);
                            m[0] = f.x; m[1] = f.y; m[2] = f.z; m[3] = f.w;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (j0 + r < L) m[r] = drop_scale(p.drop_p, mbase + j0 + r);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[jt][r] = pv[jt][r] * inv * m[r];
                }
                f32x4 o[HC];
#pragma unroll
                for (int dt = 0; dt < HC; ++dt) o[dt] = zero4();
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int dt = 0; dt < HC; ++dt) o[dt] = mfma4(vf[jt][e][dt], pv[jt][e], o[dt]);
                if (i < L) {
#pragma unroll
                    for (int dt = 0; dt < HC; ++dt) {
                        *reinterpret_cast<float4*>(tile + (r0 + i) * PQ + hc + 16 * dt + 4 * g) = f4(o[dt]);
                        st4_stream(p.av + (t0 + r0 + i) * D + hc + 16 * dt + 4 * g, o[dt]);
                    }
                }
            }
        }
    };
    switch ((2 * L + 15) / 16) {
        case 1: phase_a(IC<1>()); break;
        case 2: phase_a(IC<2>()); break;
        case 3: phase_a(IC<3>()); break;
        default: phase_a(IC<4>()); break;
    }
    __syncthreads();

    // ------------------------------------------------------------------------------------------------ phase O
    {
        const bool train = p.ao != nullptr;
        float wof[FB][4 * KC];
#pragma unroll
        for (int fb = 0; fb < FB; ++fb)
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                float t4[4];
                put4(t4, ld4(p.wo + (long)(16 * (FB * w + fb) + n) * D + 16 * c + 4 * g));
#pragma unroll
                for (int e = 0; e < 4; ++e) wof[fb][4 * c + e] = t4[e];
            }
        float4 hres[FB][R];
#pragma unroll
        for (int fb = 0; fb < FB; ++fb)
#pragma unroll
            for (int r = 0; r < R; ++r) hres[fb][r] = ld4(p.h + min(t0 + 16 * r + n, p.T - 1) * D + 16 * (FB * w + fb) + 4 * g);
        f32x4 acc[FB][R];
#pragma unroll
        for (int fb = 0; fb < FB; ++fb)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[fb][r] = zero4();
        {
            float xb[2][R][4];
#pragma unroll
            for (int r = 0; r < R; ++r) put4(xb[0][r], lds4(tile + rowc[r] * PQ + 4 * g));
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                if (c + 1 < KC) {
#pragma unroll
                    for (int r = 0; r < R; ++r) put4(xb[(c + 1) & 1][r], lds4(tile + rowc[r] * PQ + 4 * g + 16 * (c + 1)));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        acc[0][r] = mfma4(wof[0][4 * c + e], xb[c & 1][r][e], acc[0][r]);
                        acc[1][r] = mfma4(wof[1][4 * c + e], xb[c & 1][r][e], acc[1][r]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float* sh_red = xbuf;                         // [2][NW][16 R]
        float4 x[FB][R];
        float sum[R];
        float4 gam[FB], bet[FB];
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) { gam[fb] = ld4(p.gamma + 16 * (FB * w + fb) + 4 * g); bet[fb] = ld4(p.beta + 16 * (FB * w + fb) + 4 * g); }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int tok = 16 * r + n;
            const long t = t0 + tok, tc = min(t, p.T - 1);
            const bool live = tok < rows;
            float sm = 0.f;
#pragma unroll
            for (int fb = 0; fb < FB; ++fb) {
                const int f0 = 16 * (FB * w + fb) + 4 * g;
                float4 v = f4(acc[fb][r]);
                if (train && live) st4_stream(p.ao + t * D + f0, acc[fb][r]);
                if (p.drop_o.p > 0.f) {
                    const float4 m = drop_scale4(p.drop_o, (unsigned long long)tc * D + f0);
                    v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
                }
                v.x += hres[fb][r].x; v.y += hres[fb][r].y; v.z += hres[fb][r].z; v.w += hres[fb][r].w;
                x[fb][r] = v;
                sm += (v.x + v.y) + (v.z + v.w);
            }
            sm += __shfl_xor(sm, 16, 64);
            sm += __shfl_xor(sm, 32, 64);
            sum[r] = sm;
        }
        if (g == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) sh_red[w * 16 * R + 16 * r + n] = sum[r];
        }
        __syncthreads();
        float mu[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float sm = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) sm += sh_red[ww * 16 * R + 16 * r + n];
            mu[r] = sm * (1.0f / D);
            float q = 0.f;
#pragma unroll
            for (int fb = 0; fb < FB; ++fb) {
                const float dx = x[fb][r].x - mu[r], dy = x[fb][r].y - mu[r], dz = x[fb][r].z - mu[r], dw = x[fb][r].w - mu[r];
                q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            sum[r] = q;
        }
        float* sh_red2 = sh_red + NW * 16 * R;
        if (g == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) sh_red2[w * 16 * R + 16 * r + n] = sum[r];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int tok = 16 * r + n;
            const long t = t0 + tok;
            float q = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) q += sh_red2[ww * 16 * R + 16 * r + n];
            const float rs = rsqrtf(q * (1.0f / D) + p.eps);
            if (tok < rows) {
#pragma unroll
                for (int fb = 0; fb < FB; ++fb) {
                    const int f0 = 16 * (FB * w + fb) + 4 * g;
                    st4(p.h1 + t * D + f0, make_float4((x[fb][r].x - mu[r]) * rs * gam[fb].x + bet[fb].x, (x[fb][r].y - mu[r]) * rs * gam[fb].y + bet[fb].y,
                                                       (x[fb][r].z - mu[r]) * rs * gam[fb].z + bet[fb].z, (x[fb][r].w - mu[r]) * rs * gam[fb].w + bet[fb].w));
                }
                if (train && w == 0 && g == 0) { p.mean[t] = mu[r]; p.rstd[t] = rs; }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side (forward)
extern "C" int t4r_xlnet_fused_supported(int D);
static size_t attn_block_smem(int D) { return ((size_t)AB_RT * (3 * D + 4) + (size_t)(D / 16) * 16 * AB_PR) * sizeof(float); }

// 1 when the attention half of a layer of this shape runs as the one-kernel-per-direction block of this file
extern "C" int t4r_xlnet_attn_block_supported(int L, int D, int n_head) {
    if (!t4r_xlnet_fused_supported(D) || n_head <= 0 || D % n_head) return 0;
    const int dh = D / n_head;
    return L >= 1 && L <= 32 && (dh == 16 || dh == 32);
}
// sessions per workgroup: as many whole sessions as fit 80 token rows
static int attn_block_sessions(int L) { return AB_RT / L; }

// Forward of the attention half: h [T, D] -> h1 [T, D] = LayerNorm(dropout(attn_vec @ o^T) + h), saving qkv [3][T][D],
// attn_vec av [T][D], lse [B][n_head][L] and (training: ao != NULL) ao [T][D], mean / rstd [T] for the backward.
// planes: t4r_xlnet_layer_prepare's buffer (its fp32 transposes); o: the layer's o weight [D][n_head][d_head] as stored;
// kr: k_r = pos_emb @ r, [2 L][D] (kr_bstride 0) or per session [B][2 L][D] (kr_bstride 2 L D).
extern "C" int t4r_xlnet_attn_block_fwd(void* stream, const float* h, const float* planes, const float* o, const float* kr,
                                        long kr_bstride, const float* r_w_bias, const float* r_r_bias, const float* gamma,
                                        const float* beta, float* qkv, float* av, float* lse, float* ao, float* mean,
                                        float* rstd, float* h1, int B, int L, int D, int n_head, float eps, float drop_p,
                                        unsigned long long seed, unsigned long long ctr_prob, unsigned long long ctr_out,
                                        const int* key_len) {
    if (B <= 0) return 0;
    T4R_CHECK_ARG(t4r_xlnet_attn_block_supported(L, D, n_head), "xlnet_attn_block_fwd: unsupported shape (L <= 32, d_head 16 / 32, d_model 32 / 64 / 128)");
    T4R_CHECK_ARG(h && planes && o && kr && r_w_bias && r_r_bias && gamma && beta && qkv && av && lse && h1, "xlnet_attn_block_fwd: null pointer");
    const bool train = ao != nullptr;
    T4R_CHECK_ARG((mean != nullptr) == train && (rstd != nullptr) == train, "xlnet_attn_block_fwd: ao, mean, rstd go together");
    T4R_CHECK_ARG(train || drop_p == 0.f, "xlnet_attn_block_fwd: dropout needs the saved activations");
    const int dh = D / n_head, S = attn_block_sessions(L);
    AttnBlockFwd p;
    p.h = h; p.wqkvT = carve_planes_32(planes, D).QKVT; p.wo = o; p.kr = kr; p.kr_bstride = kr_bstride;
    p.rw = r_w_bias; p.rr = r_r_bias; p.gamma = gamma; p.beta = beta;
    p.qkv = qkv; p.av = av; p.lse = lse; p.ao = ao; p.mean = mean; p.rstd = rstd; p.h1 = h1; p.key_len = key_len;
    p.B = B; p.L = L; p.S = S; p.T = (long)B * L;
    p.scale = 1.0f / sqrtf((float)dh); p.eps = eps;
    p.drop_p = make_drop(drop_p, seed, ctr_prob);
    p.drop_o = make_drop(drop_p, seed, ctr_out);
#ifdef T4R_AB_STAMPS
    p.stamps = g_ab_stamps;
#endif
    hipStream_t st = (hipStream_t)stream;
    {
        // the two-workgroups-per-CU layout (d_model 128, sessions of at most 40 rows each): read per call so a test can switch it
        const char* e2 = getenv("T4R_XLNET_ATTN_BLOCK2");
        if (e2 && atoi(e2) && D == 128 && L <= AB2_RT) {
            p.S = AB2_RT / L;
            const dim3 grid2((unsigned)((B + p.S - 1) / p.S)), block2(256);
            const size_t smem2 = ((size_t)AB2_RT * (3 * D + 4) + (size_t)4 * 16 * AB_PR) * sizeof(float);
            static bool once2[2] = {false, false};
            if (dh == 32) {
                if (!once2[0]) { (void)hipFuncSetAttribute((const void*)xlnet_attn_block2_fwd_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2); once2[0] = true; }
                hipLaunchKernelGGL((xlnet_attn_block2_fwd_kernel<32>), grid2, block2, smem2, st, p);
            } else {
                if (!once2[1]) { (void)hipFuncSetAttribute((const void*)xlnet_attn_block2_fwd_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2); once2[1] = true; }
                hipLaunchKernelGGL((xlnet_attn_block2_fwd_kernel<16>), grid2, block2, smem2, st, p);
            }
            T4R_LAUNCH_CHECK();
            return 0;
        }
    }
    const dim3 grid((unsigned)((B + S - 1) / S)), block((unsigned)(D * 4));
    const size_t smem = attn_block_smem(D);
#define T4R_AB_FWD(DD, DHH)                                                                                                  \
    {                                                                                                                        \
        static bool once = false;                                                                                            \
        if (!once) { (void)hipFuncSetAttribute((const void*)xlnet_attn_block_fwd_kernel<DD, DHH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); once = true; } \
        hipLaunchKernelGGL((xlnet_attn_block_fwd_kernel<DD, DHH>), grid, block, smem, st, p);                                \
    }
    switch (D * 100 + dh) {
        case 12832: T4R_AB_FWD(128, 32) break;
        case 12816: T4R_AB_FWD(128, 16) break;
        case 6432: T4R_AB_FWD(64, 32) break;
        case 6416: T4R_AB_FWD(64, 16) break;
        case 3232: T4R_AB_FWD(32, 32) break;
        case 3216: T4R_AB_FWD(32, 16) break;
        default: t4r_set_error("xlnet_attn_block_fwd: no instantiation"); return -1;
    }
#undef T4R_AB_FWD
    T4R_LAUNCH_CHECK();
    return 0;
}

// =====================================================================================================================
// BACKWARD of the attention half: dy = d loss / d h1 -> d h (complete), and the rows the caller's weight-gradient
// products contract over (d attn_out, d q | d k | d v, d k_r); d gamma / d beta / d r_w_bias / d r_r_bias (and d k_r when it
// is shared by the sessions) as per-workgroup partial sums for one fixed-order reduction.  Autograd of the chain the
// forward kernel restates (HF modeling_xlnet.py post_attention :142-152, rel_attn_core :96-140, :251-258).
//
//   phase 1   LayerNorm backward, one wave per row at a time (xlnet_ln1_bwd_kernel's arithmetic): residual part of d h ->
//             memory, d attn_out rows -> memory (the o weight gradient) and LDS
//   phase 2   d attn_vec^T = o^T d attn_out^T: wave w owns 16 (head, d) features for all rows -> LDS tile `sv`
//   phase 3   the attention core backward, SC sessions staged at a time (q | k | v rows in LDS), one (session, head) unit
//             per wave at a time, both query blocks of a unit; every contraction on v_mfma_f32_16x16x4_f32 with the
//             permuted k-slots of the forward: the probability / d score tiles stay in the accumulator layout, which is
//             the B operand of the contractions over the keys (d q); the contractions over the queries (d v, d k, d k_r)
//             read them back transposed from a 16-row exchange buffer per wave
//   phase 4   d h += d q W_q^T + d k W_k^T + d v W_v^T: the d q / d k / d v tiles come back through LDS one at a time
//             (every wave needs all of their rows), wave w owns 16 features of d h
struct AttnBlockBwd {
    const float *dy, *ao, *h, *mean, *rstd, *gamma;   // d loss / d h1; o-projection before dropout; layer input; LayerNorm-1 statistics
    const float* woT;        // [D][D]  o^T: rows = (head, d), columns = output feature (carve_planes_32)
    const float* wqkv[3];    // W_z [D][D] as stored: rows = input feature k, columns = (head, d)
    const float* qkv;        // [3][T][D] saved by the forward
    const float* kr; long kr_bstride;
    const float *rw, *rr, *lse;
    float *dh, *dao, *dqkv;  // [T][D], [T][D], [3][T][D]
    const float* dav;        // CORE launch only: d attn_vec [T][D] (the block launch computes it in phase 2)
    float* dkr_b;            // per-session d k_r [B][2L][D] (kr_bstride > 0), else NULL
    float* part_ln;          // [grid][2 D]: d gamma | d beta partial sums
    float* part_at;          // [grid * (d_head / 16)][2 L D + 2 D]: shared d k_r | d r_w_bias | d r_r_bias partial sums
    const int* key_len;
    int B, L, S, SC;
    long T;
    float scale;
    DropCfg drop_p, drop_o;
#ifdef T4R_AB_STAMPS
    long long* stamps;
#endif
};

// CORE = true: phase 3 alone as a launch of its own (d attn_vec read from memory, one wave per head, SC sessions staged
// at a time -- LDS small enough for two workgroups per CU); the caller runs LayerNorm backward / d attn_vec before it and
// the d h product after it (t4r_xlnet_attn_core16_bwd below).
template <int D, int DH, bool CORE>
__global__ __launch_bounds__(D * 4) void xlnet_attn_block_bwd_kernel(AttnBlockBwd p) {
    constexpr int NH = D / DH, NW = CORE ? NH : D / 16, KC = D / 16, HC = DH / 16, PQ = 3 * D + 4, PV = D + 4, XR = 40;
    extern __shared__ float smem[];
    const int L = p.L;
    float* sv = smem;                                 // [AB_RT][PV]: d attn_vec of the tile's rows (CORE: [SC L][PV], of the staged sessions)
    float* xs = smem + (CORE ? L : AB_RT) * PV;      // [XR][PQ] staged q | k | v rows of SC sessions; before: d attn_out [AB_RT][PV]; after: d q / d k / d v tile
    float* skr = xs + (CORE ? L : XR) * PQ;           // CORE: [2 L][PV] k_r rows of the staged session (or the shared ones)
    float* xbuf = skr + (CORE ? 2 * L * PV : 0);      // [NW][16][AB_PR] exchange buffers
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * p.S, nb = min(p.S, p.B - b0), rows = nb * L;
    const long t0 = (long)b0 * L, TD = p.T * D;

    AB_STAMP_B(0);
    if constexpr (!CORE) {
    // the o^T rows of phase 2 are requested first: they arrive under the LayerNorm backward
    float wf[4 * KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        float t4[4];
        put4(t4, ld4(p.woT + (long)(16 * w + n) * D + 16 * c + 4 * g));
#pragma unroll
        for (int e = 0; e < 4; ++e) wf[4 * c + e] = t4[e];
    }
    // ------------------------------------------------------------------------------------------------ phase 1
        float* sd = xs;                               // [AB_RT][PV]
        float* sh_part = xbuf;                        // [NW][2][D]
        const int c0 = lane * 2;
        const bool act = c0 < D;
        float gam[2], pg[2] = {0.f, 0.f}, pb[2] = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 2; ++e) gam[e] = act ? p.gamma[c0 + e] : 0.f;
        for (int row = w; row < AB_RT; row += NW) {
            const long t = t0 + row;
            float dxa[2] = {0.f, 0.f};
            if (row < rows) {      // wave-uniform
                const float mu = p.mean[t], rs = p.rstd[t];
                float xh[2] = {0.f, 0.f}, gg[2] = {0.f, 0.f}, dyv[2] = {0.f, 0.f}, m[2] = {1.f, 1.f}, dx[2];
                float s1 = 0.f, s2 = 0.f;
                if (act) {
                    if (p.drop_o.p > 0.f) drop_scale_vec<2>(p.drop_o, (unsigned long long)t * D + c0, true, m);
                    const float2 fo = *reinterpret_cast<const float2*>(p.ao + t * D + c0);
                    const float2 hh2 = *reinterpret_cast<const float2*>(p.h + t * D + c0);
                    const float2 dd = *reinterpret_cast<const float2*>(p.dy + t * D + c0);
                    const float xv[2] = {fo.x * m[0] + hh2.x, fo.y * m[1] + hh2.y};
                    dyv[0] = dd.x; dyv[1] = dd.y;
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
                    dx[e] = act ? rs * (gg[e] - s1 - xh[e] * s2) : 0.f;
                    dxa[e] = dx[e] * m[e];
                    pg[e] += dyv[e] * xh[e];
                    pb[e] += dyv[e];
                }
                if (act) {
                    *reinterpret_cast<float2*>(p.dh + t * D + c0) = make_float2(dx[0], dx[1]);
                    *reinterpret_cast<float2*>(p.dao + t * D + c0) = make_float2(dxa[0], dxa[1]);
                }
            }
            if (act) *reinterpret_cast<float2*>(sd + row * PV + c0) = make_float2(dxa[0], dxa[1]);
        }
        if (act) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                sh_part[(w * 2 + 0) * D + c0 + e] = pg[e];
                sh_part[(w * 2 + 1) * D + c0 + e] = pb[e];
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * D; i += NW * 64) {
            float sm = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) sm += sh_part[ww * 2 * D + i];
            p.part_ln[(long)blockIdx.x * 2 * D + i] = sm;
        }
        AB_STAMP_B(1);
        // -------------------------------------------------------------------------------------------- phase 2
        // d attn_vec^T[nd][token] = sum_h o[h][nd] d ao[token][h]
        f32x4 acc[AB_R];
#pragma unroll
        for (int r = 0; r < AB_R; ++r) acc[r] = zero4();
        const float* db = sd + n * PV + 4 * g;
        float xb[2][AB_R][4];
#pragma unroll
        for (int r = 0; r < AB_R; ++r) put4(xb[0][r], lds4(db + 16 * r * PV));
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            if (c + 1 < KC) {
#pragma unroll
                for (int r = 0; r < AB_R; ++r) put4(xb[(c + 1) & 1][r], lds4(db + 16 * r * PV + 16 * (c + 1)));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int r = 0; r < AB_R; ++r) acc[r] = mfma4(wf[4 * c + e], xb[c & 1][r][e], acc[r]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < AB_R; ++r) *reinterpret_cast<float4*>(sv + (16 * r + n) * PV + 16 * w + 4 * g) = f4(acc[r]);
    AB_STAMP_B(2);
    __syncthreads();
    }
    AB_STAMP_B(3);

    // ------------------------------------------------------------------------------------------------ phase 3
    // this wave's head (all its units share it: NW is a multiple of NH) and its sums over the units
    const int hh = w % NH, hc = hh * DH;
    float acc_rw[HC][4], acc_rr[HC][4];               // column sums of d q_ac / d q_bd: feature hc + 16 dt + 4 g + r, summed over lanes n at the end
#pragma unroll
    for (int dt = 0; dt < HC; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc_rw[dt][r] = 0.f; acc_rr[dt][r] = 0.f; }
    const bool shared_kr = p.dkr_b == nullptr;
    auto phase3 = [&](auto MTc) __attribute__((always_inline)) {
        constexpr int MT = decltype(MTc)::value, JT = (MT + 1) / 2;
        float* Y = xbuf + w * 16 * AB_PR;             // [16][AB_PR]: raw^T gather, then P~ / d S of the query block by rows
        const bool aligned = (L & 3) == 0;
        f32x4 dkrT[MT][HC];                           // d k_r^T of the unit; shared k_r: summed over all of this wave's units
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int dt = 0; dt < HC; ++dt) dkrT[mt][dt] = zero4();
        // the wave's head is fixed: its bias rows / columns once
                        float rwv[4 * HC], rrv[4 * HC], rwc[HC], rrc[HC];
        #pragma unroll
                        for (int c = 0; c < HC; ++c) {
                            float t4[4];
                            put4(t4, ld4(p.rw + hc + 16 * c + 4 * g));
        #pragma unroll
                            for (int e = 0; e < 4; ++e) rwv[4 * c + e] = t4[e];
                            put4(t4, ld4(p.rr + hc + 16 * c + 4 * g));
        #pragma unroll
                            for (int e = 0; e < 4; ++e) rrv[4 * c + e] = t4[e];
                            rwc[c] = p.rw[hc + 16 * c + n];
                            rrc[c] = p.rr[hc + 16 * c + n];
                        }
        AB_STAMP_3(0, true);
        for (int s0 = 0; s0 < nb; s0 += p.SC) {
            const int sc = min(p.SC, nb - s0);
            // stage q | k | v rows of sessions s0 .. s0 + sc - 1 (coalesced).  Measured and not kept: all requests of a thread in
            // flight at once through registers, the first chunk requested before phase 1 -- the kernel is at its register limit
            // and the staging registers spilled (phase 3: 170 k -> 220 k cycles)
            if constexpr (CORE) lds_barrier();        // the previous session's unit is done with the tiles (its stores to memory need not have landed)
            else __syncthreads();                     // the previous chunk's units are done with xs (first chunk: phase 2 with sd)
            if constexpr (CORE) {
                // one session (SC = 1): q | k | v rows, d attn_vec rows and the session's k_r rows (shared k_r: once), every
                // request of a batch in flight before the first LDS store (a load -> store loop is one memory latency per turn)
                constexpr int F4 = D / 4, NB = T4R_CORE_NB;
                const int nrow = ((!shared_kr || s0 == 0) ? 6 : 4) * L;
                const long tokb = t0 + (long)s0 * L;
                const float* krsrc = p.kr + (long)(b0 + s0) * p.kr_bstride;
                for (int base = 0; base < nrow * F4; base += NB * NW * 64) {
                    float4 t[NB];
#pragma unroll
                    for (int i = 0; i < NB; ++i) {
                        const int idx = min(base + i * NW * 64 + tid, nrow * F4 - 1), R = idx / F4, c4 = (idx - R * F4) * 4;
                        const int reg = R / L, row = R - reg * L;
                        const float* src = reg < 3 ? p.qkv + reg * TD + (tokb + row) * D + c4
                                           : reg == 3 ? p.dav + (tokb + row) * D + c4 : krsrc + (long)(R - 4 * L) * D + c4;
                        t[i] = ld4(src);
                    }
#pragma unroll
                    for (int i = 0; i < NB; ++i) {
                        const int idx = base + i * NW * 64 + tid, R = idx / F4, c4 = (idx - R * F4) * 4;
                        const int reg = R / L, row = R - reg * L;
                        float* dst = reg < 3 ? xs + row * PQ + reg * D + c4 : reg == 3 ? sv + row * PV + c4 : skr + (R - 4 * L) * PV + c4;
                        if (idx < nrow * F4) *reinterpret_cast<float4*>(dst) = t[i];
                    }
                }
            } else {
            for (int idx = tid; idx < sc * L * (3 * D / 4); idx += NW * 64) {
                const int row = idx / (3 * D / 4), rem = idx - row * (3 * D / 4), z = rem / (D / 4), c4 = (rem - z * (D / 4)) * 4;
                *reinterpret_cast<float4*>(xs + row * PQ + z * D + c4) = ld4(p.qkv + z * TD + (t0 + (long)s0 * L + row) * D + c4);
            }
            }
            if constexpr (CORE) lds_barrier(); else __syncthreads();
            AB_STAMP_3(1, s0 == 0);
            for (int u = w; u < sc * NH; u += NW) {
                const int sl = u / NH, s = s0 + sl, b = b0 + s, r0 = sl * L, rv = CORE ? sl * L : s * L;       // rows in xs / in sv
                const float* krb = p.kr + (long)b * p.kr_bstride;
                const int klen = p.key_len ? p.key_len[b] : L;
                const long tok0 = t0 + (long)s * L;
                f32x4 dvT[JT][HC], dkT[JT][HC];
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int dt = 0; dt < HC; ++dt) { dvT[jt][dt] = zero4(); dkT[jt][dt] = zero4(); }
                if (!shared_kr) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int dt = 0; dt < HC; ++dt) dkrT[mt][dt] = zero4();
                }
#pragma unroll 1
                for (int it = 0; it < JT; ++it) {
                    const int i = 16 * it + n, ic = min(i, L - 1);
                    const float lrow = p.lse[((long)b * NH + hh) * L + ic];     // requested before the products it follows
                    // ---- scores: S^T, raw^T (as the forward), d P^T = v d O^T
                    float pv[JT][4], dpv[JT][4];
                    {
                        float bw[4 * HC], br[4 * HC], bo[4 * HC];
#pragma unroll
                        for (int c = 0; c < HC; ++c) {
                            float t4[4];
                            put4(t4, lds4(xs + (r0 + ic) * PQ + hc + 16 * c + 4 * g));
#pragma unroll
                            for (int e = 0; e < 4; ++e) { bw[4 * c + e] = t4[e] + rwv[4 * c + e]; br[4 * c + e] = t4[e] + rrv[4 * c + e]; }
                            put4(t4, lds4(sv + (rv + ic) * PV + hc + 16 * c + 4 * g));
#pragma unroll
                            for (int e = 0; e < 4; ++e) bo[4 * c + e] = t4[e];
                        }
                        // k_r rows from memory (L1 / L2 after the unit's first query block), requested before the products
                        // that do not need them; NOT kept across the query blocks (the kernel is at its register limit)
                        float krf[MT][4 * HC];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int c = 0; c < HC; ++c) {
                                float t4[4];
                                if constexpr (CORE) put4(t4, lds4(skr + min(16 * mt + n, 2 * L - 1) * PV + hc + 16 * c + 4 * g));
                                else put4(t4, ld4(krb + (long)min(16 * mt + n, 2 * L - 1) * D + hc + 16 * c + 4 * g));
#pragma unroll
                                for (int e = 0; e < 4; ++e) krf[mt][4 * c + e] = t4[e];
                            }
                        f32x4 sT[JT], dT[JT], rT[MT];
#pragma unroll
                        for (int jt = 0; jt < JT; ++jt) { sT[jt] = zero4(); dT[jt] = zero4(); }
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) rT[mt] = zero4();
#pragma unroll
                        for (int c = 0; c < HC; ++c) {
                            float kx[JT][4], vx[JT][4];
#pragma unroll
                            for (int jt = 0; jt < JT; ++jt) {
                                put4(kx[jt], lds4(xs + (r0 + min(16 * jt + n, L - 1)) * PQ + D + hc + 16 * c + 4 * g));
                                put4(vx[jt], lds4(xs + (r0 + min(16 * jt + n, L - 1)) * PQ + 2 * D + hc + 16 * c + 4 * g));
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
#pragma unroll
                                for (int jt = 0; jt < JT; ++jt) {
                                    sT[jt] = mfma4(kx[jt][e], bw[4 * c + e], sT[jt]);
                                    dT[jt] = mfma4(vx[jt][e], bo[4 * c + e], dT[jt]);
                                }
#pragma unroll
                                for (int mt = 0; mt < MT; ++mt) rT[mt] = mfma4(krf[mt][4 * c + e], br[4 * c + e], rT[mt]);
                            }
                        }
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<float4*>(Y + n * AB_PR + 16 * mt + 4 * g) = f4(rT[mt]);
                        wave_lds_sync();
#pragma unroll
                        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int j = 16 * jt + 4 * g + r, jj = min(j, L - 1);
                                const float sc_ = (sT[jt][r] + Y[n * AB_PR + jj + L - ic]) * p.scale;
                                const bool ok = i < L && j < L && !(j >= klen && j != i);
                                pv[jt][r] = ok ? __expf(sc_ - lrow) : 0.f;
                                dpv[jt][r] = dT[jt][r];
                            }
                        wave_lds_sync();
                    }
                    AB_STAMP_3(2, s0 == 0 && u == w && it == 0);
                    // ---- softmax backward in place: P~ = P m, d S = P (d P m - sum_j P d P m) scale
                    float ds[JT][4];
                    {
                        const unsigned long long mbase = ((unsigned long long)(b * NH + hh) * L + ic) * L;
                        float drow = 0.f;
#pragma unroll
                        for (int jt = 0; jt < JT; ++jt) {
                            const int j0 = 16 * jt + 4 * g;
                            float m[4] = {1.f, 1.f, 1.f, 1.f};
                            if (p.drop_p.p > 0.f && j0 < L) {
                                if (aligned) {
                                    const float4 f = drop_scale4(p.drop_p, mbase + j0);
                                    m[0] = f.x; m[1] = f.y; m[2] = f.z; m[3] = f.w;
                                } else {
#pragma unroll
                                    for (int r = 0; r < 4; ++r)
                                        if (j0 + r < L) m[r] = drop_scale(p.drop_p, mbase + j0 + r);
                                }
                            }
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                dpv[jt][r] *= m[r];
                                drow += pv[jt][r] * dpv[jt][r];
                                ds[jt][r] = m[r];                 // parked: the mask
                            }
                        }
                        drow += __shfl_xor(drow, 16, 64);
                        drow += __shfl_xor(drow, 32, 64);
                        // P~ rows of this query block -> Y (for d v)
#pragma unroll
                        for (int jt = 0; jt < JT; ++jt) {
                            f32x4 pd;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                pd[r] = pv[jt][r] * ds[jt][r];
                                ds[jt][r] = pv[jt][r] * (dpv[jt][r] - drow) * p.scale;
                            }
                            *reinterpret_cast<float4*>(Y + n * AB_PR + 16 * jt + 4 * g) = f4(pd);
                        }
                        wave_lds_sync();
                    }
                    AB_STAMP_3(3, s0 == 0 && u == w && it == 0);
                    // ---- d v^T[d][j] += sum_{i in block} d O[i][d] P~[i][j]:  A = a d O column, B = P~ read back by columns
                    {
                        float doc[4][HC];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int dt = 0; dt < HC; ++dt)
                                doc[e][dt] = sv[(rv + min(16 * it + 4 * g + e, L - 1)) * PV + hc + 16 * dt + n];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int jt = 0; jt < JT; ++jt) {
                                const float pb_ = Y[(4 * g + e) * AB_PR + 16 * jt + n];
#pragma unroll
                                for (int dt = 0; dt < HC; ++dt) dvT[jt][dt] = mfma4(doc[e][dt], pb_, dvT[jt][dt]);
                            }
                        wave_lds_sync();
                    }
                    AB_STAMP_3(4, s0 == 0 && u == w && it == 0);
                    // ---- d S rows -> Y;  d q^T = k^T d S^T + k_r^T d raw^T;  d k^T += (q + r_w)^T d S;  d k_r^T += (q + r_r)^T d raw
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        f32x4 t;
#pragma unroll
                        for (int r = 0; r < 4; ++r) t[r] = ds[jt][r];
                        *reinterpret_cast<float4*>(Y + n * AB_PR + 16 * jt + 4 * g) = f4(t);
                    }
                    wave_lds_sync();
                    {
                        float krc[MT][4][HC];                 // k_r by columns (d q), requested before the first term's products
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
#pragma unroll
                                for (int dt = 0; dt < HC; ++dt)
                                    krc[mt][e][dt] = CORE ? skr[min(16 * mt + 4 * g + e, 2 * L - 1) * PV + hc + 16 * dt + n]
                                                          : krb[(long)min(16 * mt + 4 * g + e, 2 * L - 1) * D + hc + 16 * dt + n];
                        f32x4 dqa[HC], dqb[HC];
#pragma unroll
                        for (int dt = 0; dt < HC; ++dt) { dqa[dt] = zero4(); dqb[dt] = zero4(); }
#pragma unroll
                        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
#pragma unroll
                                for (int dt = 0; dt < HC; ++dt) {
                                    const float kc = xs[(r0 + min(16 * jt + 4 * g + e, L - 1)) * PQ + D + hc + 16 * dt + n];
                                    dqa[dt] = mfma4(kc, ds[jt][e], dqa[dt]);
                                }
                        // d raw[i][m] = d S[i][j = m - L + i] (the relative shift transposed), own row of Y
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int j = 16 * mt + 4 * g + e - L + ic;
                                const float dr = (i < L && j >= 0 && j < L) ? Y[n * AB_PR + min(max(j, 0), L - 1)] : 0.f;
#pragma unroll
                                for (int dt = 0; dt < HC; ++dt) dqb[dt] = mfma4(krc[mt][e][dt], dr, dqb[dt]);
                            }
                        float qc[4][HC];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int dt = 0; dt < HC; ++dt)
                                qc[e][dt] = xs[(r0 + min(16 * it + 4 * g + e, L - 1)) * PQ + hc + 16 * dt + n];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int ie = 16 * it + 4 * g + e;           // the query this k-slot stands for
#pragma unroll
                            for (int jt = 0; jt < JT; ++jt) {
                                const float sb = Y[(4 * g + e) * AB_PR + 16 * jt + n];
#pragma unroll
                                for (int dt = 0; dt < HC; ++dt) dkT[jt][dt] = mfma4(qc[e][dt] + rwc[dt], sb, dkT[jt][dt]);
                            }
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) {
                                const int j = 16 * mt + n - L + min(ie, L - 1);
                                const float dr = (ie < L && j >= 0 && j < L) ? Y[(4 * g + e) * AB_PR + min(max(j, 0), L - 1)] : 0.f;
#pragma unroll
                                for (int dt = 0; dt < HC; ++dt) dkrT[mt][dt] = mfma4(qc[e][dt] + rrc[dt], dr, dkrT[mt][dt]);
                            }
                        }
                        // d q of this query block; its column sums are the bias gradients
                        if (i < L) {
#pragma unroll
                            for (int dt = 0; dt < HC; ++dt) {
                                f32x4 t;
#pragma unroll
                                for (int r = 0; r < 4; ++r) t[r] = dqa[dt][r] + dqb[dt][r];
                                st4(p.dqkv + (tok0 + i) * D + hc + 16 * dt + 4 * g, f4(t));
                            }
                        }
#pragma unroll
                        for (int dt = 0; dt < HC; ++dt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) { acc_rw[dt][r] += dqa[dt][r]; acc_rr[dt][r] += dqb[dt][r]; }
                        wave_lds_sync();
                    }
                    AB_STAMP_3(5, s0 == 0 && u == w && it == 0);
                }
                AB_STAMP_3(6, s0 == 0 && u == w);
                // ---- the unit's d k, d v rows (and d k_r of the session)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    const int j = 16 * jt + n;
                    if (j < L) {
#pragma unroll
                        for (int dt = 0; dt < HC; ++dt) {
                            st4(p.dqkv + TD + (tok0 + j) * D + hc + 16 * dt + 4 * g, f4(dkT[jt][dt]));
                            st4(p.dqkv + 2 * TD + (tok0 + j) * D + hc + 16 * dt + 4 * g, f4(dvT[jt][dt]));
                        }
                    }
                }
                if (!shared_kr) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int m = 16 * mt + n;
                        if (m < 2 * L) {
#pragma unroll
                            for (int dt = 0; dt < HC; ++dt) st4_stream(p.dkr_b + ((long)b * 2 * L + m) * D + hc + 16 * dt + 4 * g, dkrT[mt][dt]);
                        }
                    }
                }
            }
        }
        AB_STAMP_3(7, true);
        // this wave's partial sums: slot (workgroup, wave / NH), columns of its head
        float* mypart = p.part_at + ((long)blockIdx.x * (NW / NH) + w / NH) * (2L * L * D + 2 * D);
        if (shared_kr) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = 16 * mt + n;
                if (m < 2 * L) {
#pragma unroll
                    for (int dt = 0; dt < HC; ++dt) st4(mypart + (long)m * D + hc + 16 * dt + 4 * g, f4(dkrT[mt][dt]));
                }
            }
        }
#pragma unroll
        for (int dt = 0; dt < HC; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a_ = acc_rw[dt][r], b_ = acc_rr[dt][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a_ += __shfl_xor(a_, o, 64); b_ += __shfl_xor(b_, o, 64); }
                if (n == 0) {
                    mypart[2L * L * D + hc + 16 * dt + 4 * g + r] = a_;
                    mypart[2L * L * D + D + hc + 16 * dt + 4 * g + r] = b_;
                }
            }
    };
#ifdef T4R_AB_ONLY_MT
    phase3(IC<T4R_AB_ONLY_MT>());
#else
    switch ((2 * L + 15) / 16) {
        case 1: phase3(IC<1>()); break;
        case 2: phase3(IC<2>()); break;
        case 3: phase3(IC<3>()); break;
        default: phase3(IC<4>()); break;
    }
#endif

    AB_STAMP_B(4);
    // ------------------------------------------------------------------------------------------------ phase 4
    // d h[tok][k] += sum_z sum_o d z[tok][o] W_z[k][o]: wave w owns k = 16 w .. 16 w + 15
    if constexpr (!CORE) {
        f32x4 acc[AB_R];
#pragma unroll
        for (int r = 0; r < AB_R; ++r) acc[r] = zero4();
        float* dt_ = xs;                              // [AB_RT][PV]
        constexpr int NS4 = (AB_RT * (D / 4) + NW * 64 - 1) / (NW * 64);
        float4 s4[NS4];
        auto load4 = [&](int z) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NS4; ++i) {
                const int idx = min(tid + i * NW * 64, AB_RT * (D / 4) - 1);
                const int row = idx / (D / 4), c4 = (idx - row * (D / 4)) * 4;
                s4[i] = ld4(p.dqkv + z * TD + min(t0 + row, p.T - 1) * D + c4);
            }
        };
        __syncthreads();                              // everybody's d q / d k / d v rows are in memory
        load4(0);
#pragma unroll 1
        for (int z = 0; z < 3; ++z) {
            float a[4 * KC];
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                float t4[4];
                put4(t4, ld4(p.wqkv[z] + (long)(16 * w + n) * D + 16 * c + 4 * g));
#pragma unroll
                for (int e = 0; e < 4; ++e) a[4 * c + e] = t4[e];
            }
            __syncthreads();                          // the tile buffer is free
#pragma unroll
            for (int i = 0; i < NS4; ++i) {
                const int idx = tid + i * NW * 64;
                const int row = idx / (D / 4), c4 = (idx - row * (D / 4)) * 4;
                if (idx < AB_RT * (D / 4)) *reinterpret_cast<float4*>(dt_ + row * PV + c4) = s4[i];
            }
            __syncthreads();
            if (z < 2) load4(z + 1);                  // the next tile arrives under this one's products
            const float* db = dt_ + n * PV + 4 * g;
            float xb[2][AB_R][4];
#pragma unroll
            for (int r = 0; r < AB_R; ++r) put4(xb[0][r], lds4(db + 16 * r * PV));
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                if (c + 1 < KC) {
#pragma unroll
                    for (int r = 0; r < AB_R; ++r) put4(xb[(c + 1) & 1][r], lds4(db + 16 * r * PV + 16 * (c + 1)));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int r = 0; r < AB_R; ++r) acc[r] = mfma4(a[4 * c + e], xb[c & 1][r][e], acc[r]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int r = 0; r < AB_R; ++r) {
            const int tok = 16 * r + n;
            if (tok < rows) {
                float* o = p.dh + (t0 + tok) * D + 16 * w + 4 * g;
                const float4 old = ld4(o);
                st4(o, make_float4(old.x + acc[r][0], old.y + acc[r][1], old.z + acc[r][2], old.w + acc[r][3]));
            }
        }
    }
    AB_STAMP_B(5);
}

// ------------------------------------------------------------------------------------------------ host side (backward)
static size_t attn_block_bwd_smem(int D) {
    return ((size_t)AB_RT * (D + 4) + (size_t)40 * (3 * D + 4) + (size_t)(D / 16) * 16 * AB_PR) * sizeof(float);
}
extern "C" long t4r_xlnet_attn_block_bwd_part_floats(int B, int L, int D, int n_head) {
    if (!t4r_xlnet_attn_block_supported(L, D, n_head)) return 0;
    const long grid = (B + attn_block_sessions(L) - 1) / attn_block_sessions(L);
    const int dh = D / n_head;
    return grid * 2L * D + grid * (dh / 16) * (2L * L * D + 2L * D);
}

// Backward of t4r_xlnet_attn_block_fwd (same planes / weights / k_r / Philox keys).  wq, wk, wv: the layer's q, k, v weights
// [D][n_head d_head] as stored.  Overwrites dh, dao, dqkv and dkr ([B][2L][D] when kr_bstride > 0, else [2L][D]);
// ACCUMULATES d_rw, d_rr, d_gamma, d_beta.  part: t4r_xlnet_attn_block_bwd_part_floats(B, L, D, n_head) floats.
extern "C" int t4r_xlnet_attn_block_bwd(void* stream, const float* dy, const float* ao, const float* h, const float* mean,
                                        const float* rstd, const float* gamma, const float* planes, const float* wq,
                                        const float* wk, const float* wv, const float* qkv, const float* kr, long kr_bstride,
                                        const float* r_w_bias, const float* r_r_bias, const float* lse, float* dh, float* dao,
                                        float* dqkv, float* dkr, float* d_rw, float* d_rr, float* d_gamma, float* d_beta,
                                        float* part, int B, int L, int D, int n_head, float drop_p, unsigned long long seed,
                                        unsigned long long ctr_prob, unsigned long long ctr_out, const int* key_len) {
    if (B <= 0) return 0;
    T4R_CHECK_ARG(t4r_xlnet_attn_block_supported(L, D, n_head), "xlnet_attn_block_bwd: unsupported shape (L <= 32, d_head 16 / 32, d_model 32 / 64 / 128)");
    T4R_CHECK_ARG(dy && ao && h && mean && rstd && gamma && planes && wq && wk && wv && qkv && kr && r_w_bias && r_r_bias && lse &&
                      dh && dao && dqkv && dkr && d_rw && d_rr && d_gamma && d_beta && part, "xlnet_attn_block_bwd: null pointer");
    const int dhd = D / n_head, S = attn_block_sessions(L);
    const int grid_n = (B + S - 1) / S;
    AttnBlockBwd p;
    p.dy = dy; p.ao = ao; p.h = h; p.mean = mean; p.rstd = rstd; p.gamma = gamma;
    p.woT = carve_planes_32(planes, D).OT;
    p.wqkv[0] = wq; p.wqkv[1] = wk; p.wqkv[2] = wv;
    p.qkv = qkv; p.kr = kr; p.kr_bstride = kr_bstride; p.rw = r_w_bias; p.rr = r_r_bias; p.lse = lse;
    p.dh = dh; p.dao = dao; p.dqkv = dqkv; p.dav = nullptr; p.dkr_b = kr_bstride > 0 ? dkr : nullptr;
    p.part_ln = part; p.part_at = part + (long)grid_n * 2 * D;
    p.key_len = key_len; p.B = B; p.L = L; p.S = S; p.SC = 40 / L > 0 ? 40 / L : 1; p.T = (long)B * L;
    p.scale = 1.0f / sqrtf((float)dhd);
    p.drop_p = make_drop(drop_p, seed, ctr_prob);
    p.drop_o = make_drop(drop_p, seed, ctr_out);
#ifdef T4R_AB_STAMPS
    p.stamps = g_ab_stamps;
#endif
    const dim3 grid((unsigned)grid_n), block((unsigned)(D * 4));
    const size_t smem = attn_block_bwd_smem(D);
    hipStream_t st = (hipStream_t)stream;
#define T4R_AB_BWD(DD, DHH)                                                                                                  \
    {                                                                                                                        \
        static bool once = false;                                                                                            \
        if (!once) { (void)hipFuncSetAttribute((const void*)xlnet_attn_block_bwd_kernel<DD, DHH, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); once = true; } \
        hipLaunchKernelGGL((xlnet_attn_block_bwd_kernel<DD, DHH, false>), grid, block, smem, st, p);                                \
    }
    switch (D * 100 + dhd) {
        case 12832: T4R_AB_BWD(128, 32) break;
        case 12816: T4R_AB_BWD(128, 16) break;
        case 6432: T4R_AB_BWD(64, 32) break;
        case 6416: T4R_AB_BWD(64, 16) break;
        case 3232: T4R_AB_BWD(32, 32) break;
        case 3216: T4R_AB_BWD(32, 16) break;
        default: t4r_set_error("xlnet_attn_block_bwd: no instantiation"); return -1;
    }
#undef T4R_AB_BWD
    T4R_LAUNCH_CHECK();
    int rc = t4r_reduce_partials_launch(st, p.part_ln, grid_n, d_gamma, D, 1, d_beta, D, 1, nullptr, 0, 0);
    if (rc) return rc;
    return t4r_reduce_partials_launch(st, p.part_at, grid_n * (dhd / 16), kr_bstride > 0 ? nullptr : dkr, 2 * L * D, 0, d_rw, D, 1,
                                      d_rr, D, 1);
}

// ------------------------------------------------------------------------------------------------ the core alone
// Phase 3 of the kernel above as the attention-core backward of t4r_xlnet_attn_bwd (xlnet_attn.hip dispatches here when
// q | k | v and d q | d k | d v are planes of one [3][T][D] buffer, which is how the layer holds them, and
// T4R_XLNET_ATTN_CORE16=1): one wave per head, one session staged at a time, two workgroups per CU.  part: >= grid * (2 L D + 2 D) floats, grid <= min(B, 512).
int t4r_xlnet_attn_core16_ok(int L, int D, int n_head) {
    // default OFF -- measured (round 4, B 1024 / L 20 / D 128 / 4 heads, dropout 0.3): 65.6 us alone vs 61.3 us for
    // xlnet_attn_mfma_bwd_kernel, and 3.19 vs 3.09 ms per training step; read per call so that a test can switch it
    const char* e = getenv("T4R_XLNET_ATTN_CORE16");
    return e && atoi(e) && t4r_xlnet_attn_block_supported(L, D, n_head);
}
int t4r_xlnet_attn_core16_bwd(hipStream_t st, const float* qkv, const float* kr, const float* rw, const float* rr,
                              const float* lse, const float* dout, float* dqkv, float* part, float* dkr, float* d_rw,
                              float* d_rr, int B, int L, int n_head, int d_head, float scale, long kr_bstride, DropCfg drop,
                              const int* key_len) {
    const int D = n_head * d_head;
    const int S = (B + T4R_CORE_SLOTS - 1) / T4R_CORE_SLOTS, grid_n = (B + S - 1) / S;
    AttnBlockBwd p = {};
    p.qkv = qkv; p.kr = kr; p.kr_bstride = kr_bstride; p.rw = rw; p.rr = rr; p.lse = lse;
    p.dqkv = dqkv; p.dav = dout; p.dkr_b = kr_bstride > 0 ? dkr : nullptr;
    p.part_at = part;
    p.key_len = key_len; p.B = B; p.L = L; p.S = S; p.SC = 1; p.T = (long)B * L;
    p.scale = scale;
    p.drop_p = drop;
#ifdef T4R_AB_STAMPS
    p.stamps = g_ab_stamps;
#endif
    const dim3 grid((unsigned)grid_n), block((unsigned)(n_head * 64));
    const size_t smem = ((size_t)3 * L * (D + 4) + (size_t)L * (3 * D + 4) + (size_t)n_head * 16 * AB_PR) * sizeof(float);
#define T4R_AB_CORE(DD, DHH)                                                                                                 \
    {                                                                                                                        \
        static bool once = false;                                                                                            \
        if (!once) { (void)hipFuncSetAttribute((const void*)xlnet_attn_block_bwd_kernel<DD, DHH, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); once = true; } \
        hipLaunchKernelGGL((xlnet_attn_block_bwd_kernel<DD, DHH, true>), grid, block, smem, st, p);                          \
    }
    switch (D * 100 + d_head) {
        case 12832: T4R_AB_CORE(128, 32) break;
        case 12816: T4R_AB_CORE(128, 16) break;
        case 6432: T4R_AB_CORE(64, 32) break;
        case 6416: T4R_AB_CORE(64, 16) break;
        case 3232: T4R_AB_CORE(32, 32) break;
        case 3216: T4R_AB_CORE(32, 16) break;
        default: t4r_set_error("xlnet_attn_core16_bwd: no instantiation"); return -1;
    }
#undef T4R_AB_CORE
    T4R_LAUNCH_CHECK();
    return t4r_reduce_partials_launch(st, part, grid_n, kr_bstride > 0 ? nullptr : dkr, 2 * L * D, 0, d_rw, D, 1, d_rr, D, 1);
}
