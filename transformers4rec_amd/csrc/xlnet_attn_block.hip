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
// fp32 matrix time.  In one kernel the matrix pipe is the bound: measured numbers in docs/DESIGN_rounds_1_to_4.md (4.1d).
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
    const uint16_t* wqkv_h;  // HP: the same three matrices as two fp16 planes [plane][z][D][D] (LayerPlanesH::QKVT), plane stride wpl_h,
    long wpl_h;              //     positioned by the per-matrix powers of two wscale[z] (weight_scales_kernel)
    const float* wscale;
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
    DropCfg drop_in;         // p > 0 (first layer of a stack in training mode): h is the UNDROPPED input of the model; its
    float* hin;              // input dropout (HF modeling_xlnet.py:1116) is applied on load and the dropped rows are written to
                             // hin [T, D] for the backward (residual of LayerNorm 1, operand of the q | k | v weight gradients)
#ifdef T4R_AB_STAMPS
    long long* stamps;
#endif
};

template <int V>
struct IC { static constexpr int value = V; };

// HP (round 6): the q | k | v projections of phase P -- 2.0 of the launch's 3.1 GFLOP and 30.7 k of its 115 k cycles on the fp32
// matrix instruction -- on the two-way fp16 split of the feed-forward kernels instead (three v_mfma_f32_16x16x32_f16 per k-step of
// 32: a tenth of the matrix cycles; per-token and per-matrix power-of-two scales; 3-5e-6 of the largest output against fp64
// like every other large contraction of the step).  The relative-attention core and the o-projection stay exact fp32.
template <int D, int DH, bool HP>
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
        if (p.drop_in.p > 0.f) {            // workgroup-uniform
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int idx = tid + i * NW * 64;
                const int row = min(idx, AB_RT * (D / 4) - 1) / (D / 4), c4 = (min(idx, AB_RT * (D / 4) - 1) - row * (D / 4)) * 4;
                const long t = min(t0 + row, p.T - 1);
                const float4 m = drop_scale4(p.drop_in, (unsigned long long)t * D + c4);
                hstage[i].x *= m.x; hstage[i].y *= m.y; hstage[i].z *= m.z; hstage[i].w *= m.w;
                // (rows past this workgroup's S L belong to the next one, which writes them itself)
                if (idx < AB_RT * (D / 4) && row < p.S * p.L && t0 + row < p.T) *reinterpret_cast<float4*>(p.hin + t * D + c4) = hstage[i];
            }
        }
        if constexpr (HP) {
            // weight fragments of the three matrices (hi | lo planes, 16 bytes per k-step and plane), requested behind the h rows
            AFragH<D> af[3];
            const uint16_t* wrow = p.wqkv_h + (long)(16 * w + n) * D + 8 * g;
            load_a2h<D>(af[0], wrow, p.wpl_h);
            load_a2h<D>(af[1], wrow + (long)D * D, p.wpl_h);          // (the v fragments are requested behind the q product)
            // the h rows as two fp16 planes, every row positioned by its own power of two, into the v columns of the LDS tile
            // (row r: hi plane = halves [4 D, 5 D) of the row, lo plane = [5 D, 6 D): the bytes the fp32 copy used); inverse
            // scales in the exchange buffers (free until phase A)
            uint16_t* hp = reinterpret_cast<uint16_t*>(tile);
            float* sh_inv = xbuf;
            constexpr int G4 = D / 4, HPITCH = 2 * PQ;
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int idx = tid + i * NW * 64;
                const bool live = idx < AB_RT * G4;
                const int idc = live ? idx : AB_RT * G4 - 1;
                const int row = idc / G4, c4 = (idc - row * G4) * 4;
                const float4 v = hstage[i];
                float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
#pragma unroll
                for (int o = G4 / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
                const float sc = pow2_scale(m);
                if (live) {
                    uint32_t w0[2], w1[2];
                    cut2h(v.x * sc, v.y * sc, w0);
                    cut2h(v.z * sc, v.w * sc, w1);
                    uint16_t* dst = hp + (long)row * HPITCH + 4 * D + c4;
                    *reinterpret_cast<uint2*>(dst) = make_uint2(w0[0], w1[0]);
                    *reinterpret_cast<uint2*>(dst + D) = make_uint2(w0[1], w1[1]);
                    if (c4 == 0) sh_inv[row] = 1.f / sc;
                }
            }
            __syncthreads();
            AB_STAMP_P(1);
            const uint16_t* bp = hp + (long)n * HPITCH + 4 * D + 8 * g;
            float isc[AB_R];
#pragma unroll
            for (int r = 0; r < AB_R; ++r) isc[r] = sh_inv[16 * r + n];
            // q, then k: written to the tile and (write-through) to the saved q | k | v as they finish
#pragma unroll
            for (int z = 0; z < 2; ++z) {
                f32x4 acc[AB_R];
#pragma unroll
                for (int r = 0; r < AB_R; ++r) acc[r] = zero4();
                product3h<D, AB_R, HPITCH>(af[z], bp, D, acc);
                if (z == 0) load_a2h<D>(af[2], wrow + 2L * D * D, p.wpl_h);
                const float iw = 1.f / p.wscale[z];
#pragma unroll
                for (int r = 0; r < AB_R; ++r) {
                    const int tok = 16 * r + n;
                    const float sc = isc[r] * iw;                 // two exact powers of two
                    acc[r][0] *= sc; acc[r][1] *= sc; acc[r][2] *= sc; acc[r][3] *= sc;
                    *reinterpret_cast<float4*>(tile + tok * PQ + z * D + 16 * w + 4 * g) = f4(acc[r]);
                    if (tok < rows) st4_stream(p.qkv + z * TD + (t0 + tok) * D + 16 * w + 4 * g, acc[r]);
                }
            }
            AB_STAMP_P(2);
            // v: kept in its accumulators until every wave has read the planes it replaces
            f32x4 accv[AB_R];
#pragma unroll
            for (int r = 0; r < AB_R; ++r) accv[r] = zero4();
            product3h<D, AB_R, HPITCH>(af[2], bp, D, accv);
            AB_STAMP_P(3);
            __syncthreads();
            {
                const float iw = 1.f / p.wscale[2];
#pragma unroll
                for (int r = 0; r < AB_R; ++r) {
                    const int tok = 16 * r + n;
                    const float sc = isc[r] * iw;
                    accv[r][0] *= sc; accv[r][1] *= sc; accv[r][2] *= sc; accv[r][3] *= sc;
                    *reinterpret_cast<float4*>(tile + tok * PQ + 2 * D + 16 * w + 4 * g) = f4(accv[r]);
                    if (tok < rows) st4_stream(p.qkv + 2 * TD + (t0 + tok) * D + 16 * w + 4 * g, accv[r]);
                }
            }
            AB_STAMP_P(4);
            AB_STAMP_P(5);
        } else {
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
        }   // !HP
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
        if (p.drop_in.p > 0.f) {            // the residual is the DROPPED input: the same mask again (recomputed, not re-read)
#pragma unroll
            for (int r = 0; r < AB_R; ++r) {
                const float4 m = drop_scale4(p.drop_in, (unsigned long long)min(t0 + 16 * r + n, p.T - 1) * D + 16 * w + 4 * g);
                hres[r].x *= m.x; hres[r].y *= m.y; hres[r].z *= m.z; hres[r].w *= m.w;
            }
        }
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

#ifdef T4R_EXPERIMENTAL
// the measured-and-not-kept variants of this file (two-workgroups-per-CU forward, one-kernel backward, one-wave-per-head
// backward core): tools/experimental/, compiled only into the A/B variant library (tools/experimental/build_variant.sh)
#include "../../tools/experimental/xlnet_attn_block2_fwd.inc"
#endif

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
// set and cleared inside one layer call (csrc/xlnet_layer.hip, T4R_LAYER_FUSE_INPUT): the next t4r_xlnet_attn_block_fwd of this
// thread applies the model-level input dropout keyed by `ctr` to h on load and leaves the dropped rows in `hin`
static thread_local int g_ab_in_on = 0;
static thread_local unsigned long long g_ab_in_ctr = 0;
static thread_local float* g_ab_hin = nullptr;
void t4r_xlnet_attn_block_input_dropout(int on, unsigned long long ctr, float* hin) { g_ab_in_on = on; g_ab_in_ctr = ctr; g_ab_hin = hin; }
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
    const bool hp = t4r_xlnet_body_fp16x2() != 0;          // the two-way fp16 planes exist (the product's only form)
    {
        const LayerPlanesH PH_ = carve_planes_h(planes, D);
        p.wqkv_h = PH_.QKVT; p.wpl_h = 3L * D * D; p.wscale = PH_.scale + HS_Q;
    }
    p.rw = r_w_bias; p.rr = r_r_bias; p.gamma = gamma; p.beta = beta;
    p.qkv = qkv; p.av = av; p.lse = lse; p.ao = ao; p.mean = mean; p.rstd = rstd; p.h1 = h1; p.key_len = key_len;
    p.B = B; p.L = L; p.S = S; p.T = (long)B * L;
    p.scale = 1.0f / sqrtf((float)dh); p.eps = eps;
    p.drop_p = make_drop(drop_p, seed, ctr_prob);
    p.drop_o = make_drop(drop_p, seed, ctr_out);
    p.drop_in = make_drop(g_ab_in_on ? drop_p : 0.f, seed, g_ab_in_ctr);
    p.hin = g_ab_in_on ? g_ab_hin : nullptr;
    T4R_CHECK_ARG(!(g_ab_in_on && drop_p > 0.f) || p.hin, "xlnet_attn_block_fwd: the fused input dropout needs its output buffer");
#ifdef T4R_AB_STAMPS
    p.stamps = g_ab_stamps;
#endif
    hipStream_t st = (hipStream_t)stream;
#ifdef T4R_EXPERIMENTAL
    { int rc2 = 0; if (attn_block2_try_launch(p, st, B, L, D, dh, &rc2)) return rc2; }     // tools/experimental (T4R_XLNET_ATTN_BLOCK2=1)
#endif
    const dim3 grid((unsigned)((B + S - 1) / S)), block((unsigned)(D * 4));
    const size_t smem = attn_block_smem(D);
#define T4R_AB_FWD(DD, DHH)                                                                                                  \
    if (hp) {                                                                                                                \
        static T4rLdsAttr once;                                                                                            \
        t4r_ensure_dynamic_lds((const void*)xlnet_attn_block_fwd_kernel<DD, DHH, true>, smem, once);                         \
        hipLaunchKernelGGL((xlnet_attn_block_fwd_kernel<DD, DHH, true>), grid, block, smem, st, p);                          \
    } else {                                                                                                                 \
        static T4rLdsAttr once;                                                                                            \
        t4r_ensure_dynamic_lds((const void*)xlnet_attn_block_fwd_kernel<DD, DHH, false>, smem, once);                        \
        hipLaunchKernelGGL((xlnet_attn_block_fwd_kernel<DD, DHH, false>), grid, block, smem, st, p);                         \
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

#ifdef T4R_EXPERIMENTAL
#include "../../tools/experimental/xlnet_attn_block_bwd.inc"
#endif
