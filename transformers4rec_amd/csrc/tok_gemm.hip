// Token-stationary GEMM for the transformer body:  C[M, N] = alpha * A[M, K] @ op(B) (+ epilogue), M = tokens (tens of
// thousands), K and N = d_model / d_inner (32 .. 512) -- the q/k/v/r/o projections and the feed-forward of
// HF modeling_xlnet.py:253-259,145,297-305 (through transformers4rec/torch/block/transformer.py:179-199), the
// GPT-2 / BERT projections of the same shapes, and their d X contractions.
//
// fp32-accurate on the BF16 matrix cores (the exact three-way split of gemm_kernel.h PREC 1), with the cutting hoisted
// the way csrc/head_split.hip does it for the head: on these shapes the general kernel re-cuts every 64-row A tile
// once per 64 columns of N and every B tile once per 64 rows of M (320 times at 20 480 tokens), and its VALU work
// equals its matrix-core time.  Here
//   * a wave owns 32 token rows for its whole life: their A fragments (the full K extent, or 128-wide chunks of it)
//     are cut ONCE, straight into MFMA operand registers -- each lane reads a contiguous half row, nothing goes
//     through LDS;
//   * the weight (small: at most 512 x 128) streams through LDS as 32-column plane blocks cut by the workgroup while
//     the previous block is multiplied (16 values per thread per block -- a sixth of the matrix-core time);
//   * fragments come out of LDS as one ds_read_b128 per plane for BOTH weight orientations (the cut transposes).
// Grid: (token tiles of 128, column groups of NBW x 32 columns, batch).  Epilogues and dropout-mask indexing are the
// general kernel's (same C fragment layout: lane = column, registers = rows), so masks and results line up with it.
#include "gemm_kernel.h"
#include <atomic>

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

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
__device__ __forceinline__ f32x16 mfma6(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 acc) {
    acc = mfma_bf16(a[1], b[1], acc);
    acc = mfma_bf16(a[2], b[0], acc);
    acc = mfma_bf16(a[0], b[2], acc);
    acc = mfma_bf16(a[1], b[0], acc);
    acc = mfma_bf16(a[0], b[1], acc);
    acc = mfma_bf16(a[0], b[0], acc);
    return acc;
}

// NBK: width of one K chunk in units of 32 (K <= 128: the whole K; K > 128: 128-wide chunks, KC of them)
// TB : B is [N][K] (k contiguous) / [K][N] (n contiguous);  NBW: 32-column blocks per workgroup
template <int NBK, bool TB, int NBW, bool EDROP>
__global__ __launch_bounds__(256) void tok_gemm_kernel(GemmParams p, int KC) {
    constexpr int KW = 32 * NBK, KS = 2 * NBK, CH = 4 * NBK;
    constexpr int BLK = 3 * CH * 32;                 // u32x4 per weight plane block
    constexpr int ITEMS = CH * 32;                   // (chunk, column) pieces of one block
    constexpr int NIT = (ITEMS + 255) / 256;
    __shared__ u32x4 lds[2][BLK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, khalf = lane >> 5;
    const int m0 = blockIdx.x * 128, n0 = blockIdx.y * (NBW * 32), batch = blockIdx.z;
    const float* A = p.A + batch * p.sA;
    const float* B = p.B + batch * p.sB;
    float* C = p.C + batch * p.sC;
    const float* arow = A + (long)min(m0 + 32 * wave + l32, p.M - 1) * p.lda + (KW / 2) * khalf;

    // the k-slot (khalf, e) of MFMA step s holds physical k = chunk_base + (KW / 2) khalf + 8 s + e: a lane reads one
    // contiguous half row; the weight chunk feeding it is c = KS khalf + s
    u32x4 Af[KS][3];
    auto a_load = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float4 u = *reinterpret_cast<const float4*>(arow + kc * KW + 8 * s);
            const float4 v = *reinterpret_cast<const float4*>(arow + kc * KW + 8 * s + 4);
            const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
            split8(x, Af[s]);
        }
    };
    // weight block t = kc * NBW + jj: columns n0 + 32 jj .., k range [KW kc, KW kc + KW): raw values in registers
    float wst[NIT][8];
    auto w_load = [&](int t) __attribute__((always_inline)) {
        const int kc = t / NBW, jj = t % NBW;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = min(it * 256 + tid, ITEMS - 1), r = idx & 31, c = idx >> 5;
            const int n = n0 + 32 * jj + r, k = kc * KW + 8 * c;
            if (TB) {
                const float4 u = *reinterpret_cast<const float4*>(B + (long)n * p.ldb + k);
                const float4 v = *reinterpret_cast<const float4*>(B + (long)n * p.ldb + k + 4);
                wst[it][0] = u.x; wst[it][1] = u.y; wst[it][2] = u.z; wst[it][3] = u.w;
                wst[it][4] = v.x; wst[it][5] = v.y; wst[it][6] = v.z; wst[it][7] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) wst[it][e] = B[(long)(k + e) * p.ldb + n];
            }
        }
    };
    auto w_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * 256 + tid, r = idx & 31, c = idx >> 5;
            u32x4 w[3];
            split8(wst[it], w);
            if (ITEMS % 256 == 0 || idx < ITEMS) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) lds[buf][(pl * CH + c) * 32 + r] = w[pl];
            }
        }
    };

    f32x16 acc[NBW];
#pragma unroll
    for (int j = 0; j < NBW; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int n_blocks = KC * NBW;
    w_load(0);
    a_load(0);
    w_store(0);
    __syncthreads();
    for (int kc = 0; kc < KC; ++kc) {
        if (kc > 0) a_load(kc);
#pragma unroll
        for (int jj = 0; jj < NBW; ++jj) {
            const int t = kc * NBW + jj, buf = t & 1;
            w_load(min(t + 1, n_blocks - 1));           // unconditional: a branch here would cost the register allocation
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                u32x4 bf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bf[pl] = lds[buf][(pl * CH + KS * khalf + s) * 32 + l32];
                acc[jj] = mfma6(Af[s], bf, acc[jj]);
            }
            w_store(buf ^ 1);
            __syncthreads();
        }
    }

    // epilogue: the general kernel's (gemm_kernel.h), on the same fragment layout
    const float alpha = p.alpha;
    const int row0 = m0 + 32 * wave + 4 * khalf;
    const bool rows_full = m0 + 128 <= p.M;
    auto for_each_out = [&](auto fn) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NBW; ++j) {
            const int col = n0 + 32 * j + l32;
            float* c0 = C + (long)row0 * p.ldc + col;
            float dm[16];
            if (EDROP) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float m4[4];
                    drop_scale_quad(p.drop, (unsigned long long)(row0 + 8 * g), (unsigned long long)p.N, col, m4);
                    dm[4 * g] = m4[0]; dm[4 * g + 1] = m4[1]; dm[4 * g + 2] = m4[2]; dm[4 * g + 3] = m4[3];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                if (!rows_full && row0 + dr >= p.M) continue;
                fn(c0 + (long)dr * p.ldc, row0 + dr, col, alpha * acc[j][r], EDROP ? dm[r] : 1.f);
            }
        }
    };
    if (p.epilogue == EPI_NONE) {
        if (p.accumulate) for_each_out([&](float* cp, int, int, float v, float) __attribute__((always_inline)) { *cp += v; });
        else for_each_out([&](float* cp, int, int, float v, float) __attribute__((always_inline)) { *cp = v; });
    } else {
        const int mode = p.epilogue;
        const bool acc_c = p.accumulate;
        for_each_out([&](float* cp, int row, int col, float v, float dmask) __attribute__((always_inline)) {
            const float bv = p.bias ? p.bias[col] : 0.f;
            if (mode == EPI_BIAS) {
                v += bv;
            } else if (mode == EPI_BIAS_GELU) {
                v += bv;
                if (p.aux) p.aux[(long)row * p.ldaux + col] = v;
                v = gelu_erf(v);
                if (EDROP) v *= dmask;
            } else if (mode == EPI_BIAS_RELU) {
                v = fmaxf(v + bv, 0.f);
            } else if (mode == EPI_BIAS_RESID) {
                v += bv;
                if (EDROP) v *= dmask;
                v += p.aux[(long)row * p.ldaux + col];
            }
            if (acc_c) v += *cp;
            *cp = v;
        });
    }
}

template <int NBK, bool TB, int NBW>
int launch_drop(const GemmParams& p, int batch, int KC, hipStream_t stream) {
    dim3 grid((p.M + 127) / 128, p.N / (32 * NBW), batch);
    const bool edrop = p.drop.p > 0.f && (p.epilogue == EPI_BIAS_GELU || p.epilogue == EPI_BIAS_RESID);
    if (edrop) hipLaunchKernelGGL((tok_gemm_kernel<NBK, TB, NBW, true>), grid, dim3(256), 0, stream, p, KC);
    else hipLaunchKernelGGL((tok_gemm_kernel<NBK, TB, NBW, false>), grid, dim3(256), 0, stream, p, KC);
    T4R_LAUNCH_CHECK();
    return 1;
}
template <int NBK, bool TB>
int launch_nbw(const GemmParams& p, int batch, int KC, hipStream_t stream) {
    // column blocks per workgroup: as many as keep >= ~2 workgroups per CU in the grid (each workgroup cuts its 128 token
    // rows once, whatever its share of N), at most 4 (accumulator registers)
    const long tiles = (long)((p.M + 127) / 128) * batch;
    const int nb = p.N / 32;
    static int force = -1;
    if (force < 0) { const char* e = t4r_exp_getenv("T4R_TOK_NBW"); force = e ? atoi(e) : 0; }
    int nbw = 1;
    for (int c : {4, 2}) {
        if (nb % c == 0 && tiles * (nb / c) >= 480) { nbw = c; break; }
    }
    if (force == 1 || force == 2 || force == 4) { if (nb % force == 0) nbw = force; }
    if (nbw == 4) return launch_drop<NBK, TB, 4>(p, batch, KC, stream);
    if (nbw == 2) return launch_drop<NBK, TB, 2>(p, batch, KC, stream);
    return launch_drop<NBK, TB, 1>(p, batch, KC, stream);
}

}  // namespace

// 1: launched;  0: shape / operands not covered (the caller takes the general kernel);  < 0: launch error.
// Covered: op(A) = A, N % 32 == 0, K in {32, 64, 96, 128} or a multiple of 128, 16-byte loadable A (and B when transB),
// no split-K, no softmax-gradient / rank variants.
// Where it is used: measured on MI355X (tools/tok_gemm_bench.py, profiles/r02_*): at 20 480 tokens (BASELINE configs[1]) these
// products are bound by launch + memory latency in EITHER kernel (14 us for a 0.67 GFLOP projection, 31 us for the 2.7 GFLOP
// feed-forward ones) and the general kernel's smaller register footprint wins inside the step; from ~50 k tokens on
// (configs[3] / [4]: 51 200 x 256 .. 102 400 x 512 x 2048) this kernel is 4-11 % faster for K <= 512.  Default: rows >= 32 768 and
// K <= 512; T4R_TOK_GEMM_MIN_M / t4r_set_tok_gemm_min_rows() move the row threshold (the tests run it at 512), T4R_TOK_GEMM=0
// turns it off.
static std::atomic<int> g_min_rows{-1};
extern "C" void t4r_set_tok_gemm_min_rows(int rows) { g_min_rows.store(rows < 0 ? -1 : rows); }
extern "C" int t4r_get_tok_gemm_min_rows(void) {
    int m = g_min_rows.load();
    if (m < 0) {
        const char* e = t4r_exp_getenv("T4R_TOK_GEMM_MIN_M");
        m = e ? atoi(e) : 32768;
        g_min_rows.store(m);
    }
    return m;
}
int t4r_tok_gemm_try(const GemmParams& p, int batch, int ta, int tb, hipStream_t stream) {
    static int on = -1;
    if (on < 0) { const char* e = t4r_exp_getenv("T4R_TOK_GEMM"); on = e ? atoi(e) : 1; }
    if (!on || ta || p.sg_lse || p.rk_thr || p.splitk > 1) return 0;
    // below the row threshold only the wide, short products without an epilogue (K <= 128, N >= 512: d ff = d ffout @ W2
    // at 20 480 tokens) -- the one body shape of configs[1] where this kernel was faster INSIDE the step (56 vs 81 us)
    static int wide = -1;
    if (wide < 0) { const char* e = t4r_exp_getenv("T4R_TOK_GEMM_WIDE"); wide = e ? atoi(e) : 1; }
    const bool wide_short = wide && p.K <= 128 && p.N >= 512 && p.epilogue == EPI_NONE && p.M >= 8192;
    if ((p.M < t4r_get_tok_gemm_min_rows() && !wide_short) || p.N % 32 || p.N < 32 || p.K % 32 || p.K < 32 || p.K > 512) return 0;
    if (p.K > 128 && p.K % 128) return 0;
    if (!p.vecA || (tb && !p.vecB)) return 0;
    const int nbk = p.K > 128 ? 4 : p.K / 32, KC = p.K > 128 ? p.K / 128 : 1;
    switch (nbk) {
        case 1: return tb ? launch_nbw<1, true>(p, batch, KC, stream) : launch_nbw<1, false>(p, batch, KC, stream);
        case 2: return tb ? launch_nbw<2, true>(p, batch, KC, stream) : launch_nbw<2, false>(p, batch, KC, stream);
        case 3: return tb ? launch_nbw<3, true>(p, batch, KC, stream) : launch_nbw<3, false>(p, batch, KC, stream);
        default: return tb ? launch_nbw<4, true>(p, batch, KC, stream) : launch_nbw<4, false>(p, batch, KC, stream);
    }
}
