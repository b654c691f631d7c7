// fp32 GEMM on the CDNA4 matrix cores:  C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] (+ epilogue)
//
// Used for every dense contraction of the session-sequence hot path (exact-fp32 parity with
// the reference's CPU path, which computes in fp32):
//   XLNet q/k/v/o/r projections   (HF modeling_xlnet.py:253-259,145)   einsum("ibh,hnd->ibnd")
//   XLNet feed-forward            (HF modeling_xlnet.py:297-305)
//   projection MLP / task block   (transformers4rec/torch/block/mlp.py:133-135)
//   next-item logits X @ W^T      (transformers4rec/torch/model/prediction_task.py:664)
// and all of their dgrad / wgrad contractions.
//
// Design (gfx950): v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD, 157 TF chip peak).
// 256-thread workgroup = 4 waves as 2(M) x 2(N); block tile BM x BN x BK, wave tile
// (BM/2) x (BN/2) = WMxWN MFMA tiles of 32x32.  Each operand keeps in LDS the orientation it has
// in HBM, so it is staged with plain 16-byte global loads and 16-byte LDS stores:
//   k-contiguous operand (A of N?, B of ?T):  image S[m][k], row pitch BK, 16-byte chunks
//                                              XOR-swizzled by the row (bank-conflict free for
//                                              the stage stores AND the fragment reads)
//   m-contiguous operand (A of T?, B of ?N):  image S[k][m], row pitch BM+4 floats
// The k-slots of the MFMA are permuted so that a lane reads CONTIGUOUS k from an S[m][k] image:
// MFMA step s of a k-tile takes physical k = (lane>>5)*(BK/2) + s for both operands (any
// bijection k <-> (step, half) is a valid contraction order).  A lane's fragments for four steps
// are then one ds_read_b128 instead of four half-rate ds_read_b32.  S[k][m] images are read with
// ds_read_b32.
// Software pipeline, three stages deep (see the k-loop): the tile being multiplied has its MFMA
// fragments in registers, the next one is moving from its register stage into LDS, the one after
// that is in flight from memory; staging loads are branch-free (clamped addresses, k-tail zeroed
// on the way to LDS).  Measured end of round 1 (tools/gemm_bench.py, uniform data): square 4096^3
// 97-116 TF/s, head logits 84-91, head dW / dX 95-105, K = 128 layer GEMMs 40-75; a pure-MFMA loop
// reaches 156 (tools/mfma_peak.hip); tools/gemm_ablate.hip shows where the difference goes
// (global loads ~28 %, the logits' 1.1 GB epilogue ~20 %, data-dependent clocks ~20 %).
// Split-K (gridDim.z) accumulates with hardware fp32 atomics into a zeroed / accumulating C:
// this is how every weight gradient (K = tokens) and the head's dX (K = vocabulary) get
// enough workgroups to fill 256 CUs.
#pragma once
#include "t4r_common.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_GELU = 2, EPI_BIAS_RELU = 3, EPI_BIAS_RESID = 4 };

struct GemmParams {
    int M, N, K;
    const float* A; long lda;   // TA=0: A[M][lda] (k contiguous)   TA=1: A[K][lda] (m contiguous)
    const float* B; long ldb;   // TB=0: B[K][ldb] (n contiguous)   TB=1: B[N][ldb] (k contiguous)
    float* C; long ldc;
    const float* bias;          // [N] or null
    float* aux; long ldaux;     // EPI_BIAS_GELU: pre-activation (x + bias) written here
    float alpha;
    int epilogue;
    int splitk;                 // >1: atomic accumulate alpha*partial into C (epilogue must be NONE) ...
    const float* amaxA;         // PREC 4: n_amax per-workgroup maxima of |A| and of |B| (device arrays written by the producers
    const float* amaxB;         //   of the operands): the two-way fp16 split positions both operands by powers of two of their max
    int n_amax;                 //   n_amax values of A, n_amax_b of B (the producers of the two operands may have run different grids)
    int n_amax_b;
    float* part;                // ... or, non-null: split z (= blockIdx.z, batch-major) stores its partial tile to
                                // part + z * M * ldc (C's layout); a reduction pass adds them in split order (gemm_f32.hip)
    int accumulate;             // splitk==1 only: C += result instead of C = result
    long sA, sB, sC;            // batch strides in elements (gridDim.z = batch * splitk)
    int vecA, vecB;             // 16-byte vector loads legal for this operand
    int xcd_order;              // XCD-aware tile order (see kernel)
    DropCfg drop;               // EPI_BIAS_GELU only: C = dropout(gelu(x + bias)), mask index row*N + col
    // A-operand transform: A holds LOGITS [rows, V]; the GEMM consumes the softmax-CE gradient
    //   a = (*gout / n_rows) * (exp(x - lse[row]) - (1-eps)*[col == y[row]] - eps/V)
    // computed on the fly while staging the tile (fuses CrossEntropyLoss backward into the
    // head's dX / dW contractions: the [N, V] gradient never goes to HBM).
    const float* sg_lse;        // null => plain A
    const long* sg_labels;
    const float* sg_gout;       // device scalar (d loss), may be null (=1)
    int sg_rows, sg_V;
    int sg_yoff;                // column of the FIRST logit of A in the vocabulary (chunk-streamed head): label - sg_yoff is A's column
    float sg_smooth;
    // rank-of-target epilogue (FEAT bit 2; fused eval head): nothing is stored; for every output row
    // the workgroup counts the columns that beat the row's target score,
    //   rk_count[row] += #{col < N : v > thr[row]  or  (v == thr[row] and col < label[row])},
    // i.e. the 0-based rank of the target under "ties go to the lower index" (the top-k convention).
    const float* rk_thr;        // null => normal epilogue
    const long* rk_label;
    int* rk_count;
};

// softmax-gradient transform of four consecutive columns col0..col0+3 of one logits row.
// Branch-free on purpose (the k-loop schedules it under MFMAs inside ONE basic block); elements
// outside the operand are transformed too and zeroed afterwards by mask4.
__device__ __forceinline__ float4 softmax_grad4(float4 v, float l, int y, int col0, float g,
                                                 const GemmParams& p) {
    const float sub = p.sg_smooth / p.sg_V;
    const float hit = g * (1.f - p.sg_smooth);
    const int d = y - col0;     // 0..3 when the label column is one of the four
    v.x = g * (__expf(v.x - l) - sub) - (d == 0 ? hit : 0.f);
    v.y = g * (__expf(v.y - l) - sub) - (d == 1 ? hit : 0.f);
    v.z = g * (__expf(v.z - l) - sub) - (d == 2 ? hit : 0.f);
    v.w = g * (__expf(v.w - l) - sub) - (d == 3 ? hit : 0.f);
    return v;
}

// Edge handling without branches: every staging load reads a LEGAL address (row and column
// clamped into the operand) and the out-of-range elements are zeroed when the stage moves to LDS.
// A guarded "v = 0; if (in range) v = load" form makes the compiler wait for the outstanding
// loads before it may overwrite a component (s_waitcnt vmcnt(0) in the middle of the prefetch).
// rowp: start of an in-range row; c: first column (multiple of 4); lim: columns of the row.
// vec: 16-byte loads legal (base and row pitch multiples of 4 floats; the pitch then covers
// ceil4(lim), so the partial last float4 of a row stays inside the row).
template <bool VEC>
__device__ __forceinline__ float4 ld4_clamped(const float* rowp, int c, int lim) {
    if (VEC) return *reinterpret_cast<const float4*>(rowp + min(c, (lim - 1) & ~3));
    const int l = lim - 1;
    return make_float4(rowp[min(c, l)], rowp[min(c + 1, l)], rowp[min(c + 2, l)], rowp[min(c + 3, l)]);
}
// Pins the point where a staged register is first consumed: nothing computed from it (edge masks,
// the softmax-gradient transform) may be hoisted above this statement -- the compiler otherwise
// moves such pure VALU work up to the load and waits for the load there (s_waitcnt vmcnt(0) in the
// middle of the prefetch distance).
__device__ __forceinline__ void pin4(float4& v) {
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}
__device__ __forceinline__ float4 mask4(float4 v, int valid) {   // keep the first `valid` elements
    v.x = valid > 0 ? v.x : 0.f; v.y = valid > 1 ? v.y : 0.f;
    v.z = valid > 2 ? v.z : 0.f; v.w = valid > 3 ? v.w : 0.f;
    return v;
}

// FEAT bit 0: softmax-gradient A operand ; bit 1: dropout in the epilogue.  Compile-time so that the
// plain GEMM does not carry the Philox / exp code (measured: +12 % step time when it did).
// VEC: both operands can be staged with 16-byte loads (decided by the host from pointers / pitches);
// the scalar-load variant is its own instantiation so that it does not set the register budget.
// PREC: 0 = fp32 operands on the fp32 matrix cores (v_mfma_f32_32x32x2_f32) -- everything above;
//       1 = fp32-accurate products on the BF16 matrix cores: every operand is cut by truncation into three bf16
//           pieces (x = hi + mid + lo exactly) while its tile is staged, and the six largest partial products
//           (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid) are accumulated in fp32 by v_mfma_f32_32x32x16_bf16:
//           192 matrix-pipe cycles per 32x32x16 instead of 512 (error = that of fp32 arithmetic, tools/gemm_bf16x6.hip);
//       2 / 3 = mixed precision as the reference's AMP (trainer.py:363-367): operands rounded (RNE) to bf16 / fp16
//           on the way to LDS, fp32 accumulation, fp32 output -- one matrix-core product.
// PREC != 0 uses BK = 32 (two K = 16 MFMA steps per k-tile) and its own LDS images (see below); the staging
// loads, the pipeline, the split-K logic and every epilogue are shared with the fp32 path.
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// two fp32 values -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32; low half = first value)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_bf16_rne(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// two consecutive-k fp32 values -> one 32-bit word per plane (low half = first value).
// PREC 1, the exact three-way split x = hi + mid + lo: every cut ROUNDS TO NEAREST (hi = bf16(x),
// mid = bf16(x - hi), lo = x - hi - mid, each residual exact in fp32 and lo exactly a bf16), so the three dropped
// partial products (mid.lo, lo.mid, lo.lo, < 2^-24 |a||b|) carry no systematic sign -- with truncation cuts every
// piece has the sign of x and the dropped terms add up over long reductions (seen as a 1.8e-5 drift of the
// head's sum_v dW[v, :] checksum over 100 k rows).  Same instruction count as the truncation form.
// PREC 4, the two-way fp16 split of an operand already positioned by a power of two (max |.| in [2^13, 2^14)):
// hi = fp16(x), lo = fp16(x - hi), both round to nearest; three products hi.hi + hi.lo + lo.hi (csrc/head_split.hip).
template <int PREC>
__device__ __forceinline__ void cvt_pair(float a, float b, uint32_t (&w)[PREC == 1 ? 3 : (PREC == 4 ? 2 : 1)]) {
    if constexpr (PREC == 4) {
        // (vector conversions + fma(hi, -1, x): five vector instructions per pair instead of eight, same roundings)
        typedef float f2v_t __attribute__((ext_vector_type(2)));
        const f2v_t ab = {a, b};
        const half2_t h = __builtin_convertvector(ab, half2_t);
        const f2v_t lab = {__builtin_fmaf((float)h[0], -1.0f, a), __builtin_fmaf((float)h[1], -1.0f, b)};
        const half2_t l = __builtin_convertvector(lab, half2_t);
        w[0] = __builtin_bit_cast(uint32_t, h);
        w[1] = __builtin_bit_cast(uint32_t, l);
    } else if constexpr (PREC == 1) {
#ifdef T4R_SPLIT_TRUNC      // A/B build only: truncation cuts (biased, see above)
        const uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
        const float ta = a - __uint_as_float(ua & 0xffff0000u), tb = b - __uint_as_float(ub & 0xffff0000u);
        const uint32_t va = __float_as_uint(ta), vb = __float_as_uint(tb);
        const float qa = ta - __uint_as_float(va & 0xffff0000u), qb = tb - __uint_as_float(vb & 0xffff0000u);
        w[0] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
        w[1] = __builtin_amdgcn_perm(vb, va, 0x07060302u);
        w[2] = __builtin_amdgcn_perm(__float_as_uint(qb), __float_as_uint(qa), 0x07060302u);
        return;
#endif
        w[0] = pk_bf16_rne(a, b);
        const float ra = a - __uint_as_float(w[0] << 16), rb = b - __uint_as_float(w[0] & 0xffff0000u);
        w[1] = pk_bf16_rne(ra, rb);
        const float sa = ra - __uint_as_float(w[1] << 16), sb = rb - __uint_as_float(w[1] & 0xffff0000u);
        w[2] = pk_bf16_rne(sa, sb);
    } else if constexpr (PREC == 2) {                             // bf16, round to nearest even
        w[0] = pk_bf16_rne(a, b);
    } else {                                                      // fp16, round to nearest even
        const half2_t h = {(_Float16)a, (_Float16)b};
        w[0] = __builtin_bit_cast(uint32_t, h);
    }
}
template <int PREC>
__device__ __forceinline__ f32x16 mfma_half(uint4 a, uint4 b, f32x16 c) {
    if constexpr (PREC == 3 || PREC == 4)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// LDS footprint (bytes) of one workgroup: two buffers of the A and B operand images
template <int BM, int BN, int BK, bool TA, bool TB, int PREC>
constexpr size_t gemm_lds_bytes() {
    if (PREC == 0) return (size_t)2 * ((!TA ? BM * BK : BK * (BM + 4)) + (TB ? BN * BK : BK * (BN + 4))) * 4;
    constexpr int npl = PREC == 1 ? 3 : (PREC == 4 ? 2 : 1);
    constexpr int pa = (BK / 8) * (!TA ? BM * 4 + 16 : 4 * (BM + 4)), pb = (BK / 8) * (TB ? BN * 4 + 16 : 4 * (BN + 4));
    return (size_t)2 * npl * (pa + pb) * 4;
}

template <int BM, int BN, int BK, bool TA, bool TB, int FEAT, bool VEC, int PREC = 0>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
    constexpr bool SG = (FEAT & 1) != 0, EDROP = (FEAT & 2) != 0, RANK = (FEAT & 4) != 0;
    constexpr int WM = BM / 64, WN = BN / 64;          // MFMA tiles per wave per dim
    constexpr bool A_MK = !TA, B_MK = TB;              // operand image is S[m][k] (k contiguous)
    // S[m][k] images: row pitch BK (no padding) with the 16-byte chunk index XOR-swizzled by the row,
    // chunk' = chunk ^ ((row / (64/BK)) % (BK/4)).  A padded pitch cannot serve both sides: the
    // 16-byte stage stores (8 lanes = 2 rows of 16 floats, banks mod 32) need the pitch = 16 mod 32,
    // the ds_read_b128 fragment reads (16 lanes = 16 rows, banks mod 64) need pitch/4 odd; with
    // pitch BK+4 rocprofv3 counted SQ_LDS_BANK_CONFLICT = 1/3 of SQ_LDS_IDX_ACTIVE on NT GEMMs.
    // S[k][m] images keep the padded pitch BM+4 (conflict free for their access patterns).
    constexpr int LDA_S = A_MK ? BK : BM + 4;          // LDS row pitch (floats)
    constexpr int LDB_S = B_MK ? BK : BN + 4;
    auto swz = [](int row, int chunk) { return (chunk ^ ((row / (64 / BK)) % (BK / 4))) * 4; };
    constexpr bool HALF = PREC != 0;                   // bf16 / fp16 matrix cores
    constexpr int NPL = PREC == 1 ? 3 : (PREC == 4 ? 2 : 1);   // operand planes: hi | mid | lo (bf16), hi | lo (fp16 split), or one
    static_assert(!HALF || BK == 32, "the half-precision operand images are built for BK = 32");
    // half-precision images, per plane, in 32-bit words holding two consecutive k (even k in the low half):
    //   k-contiguous operand: [chunk = k/8][row][4 words]; chunk pitch ROWS*4 + 16 words -- the 8-byte stage
    //     stores of a 32-lane group (4 rows x 8 half-chunks) and the 16-byte fragment reads (16 consecutive
    //     rows of one chunk) both touch every bank once;
    //   m-contiguous operand: [chunk][word = (k%8)/2][row]; line pitch ROWS + 4 words -- a thread stages NA4
    //     CONSECUTIVE k of four columns (km_map below), i.e. whole words: one 16-byte store per word line, and a
    //     fragment is four conflict-free ds_read_b32 (consecutive rows).
    constexpr int HA_PITCH = A_MK ? BM * 4 + 16 : BM + 4, HB_PITCH = B_MK ? BN * 4 + 16 : BN + 4;
    constexpr int HA_PLANE = (BK / 8) * (A_MK ? 1 : 4) * HA_PITCH, HB_PLANE = (BK / 8) * (B_MK ? 1 : 4) * HB_PITCH;
    constexpr int A_SZ = HALF ? NPL * HA_PLANE : (A_MK ? BM * LDA_S : BK * LDA_S);   // 32-bit words per buffer
    constexpr int B_SZ = HALF ? NPL * HB_PLANE : (B_MK ? BN * LDB_S : BK * LDB_S);
    static_assert((size_t)2 * (A_SZ + B_SZ) * 4 == gemm_lds_bytes<BM, BN, BK, TA, TB, PREC>(), "LDS size formula");
    constexpr int KH = BK / 2;                         // k-slots per lane half
    constexpr int NA4 = BM * BK / 4 / 256;             // float4 per thread per tile
    constexpr int NB4 = BN * BK / 4 / 256;
    static_assert(NA4 >= 1 && NB4 >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                   // [2][A_SZ]
    float* Bs = smem + 2 * A_SZ;                        // [2][B_SZ]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // element of staging load r of this thread in an m-contiguous ("S[k][m]") operand tile -> (k row, first of 4
    // columns).  fp32 path: float4 number tid + 256 r, row-major over (k, m/4).  Half path: n4 CONSECUTIVE k rows
    // of the same four columns, so that both halves of every 32-bit word (k, k+1) come from one thread.
    auto km_map = [&](int r, int rows4, int n4, int& k, int& m4) __attribute__((always_inline)) {
        if (HALF) { k = (tid / rows4) * n4 + r; m4 = (tid % rows4) * 4; }
        else { const int idx = tid + r * 256; k = idx / rows4; m4 = (idx % rows4) * 4; }
    };
    // XCD-aware tile order.  Workgroup b runs on XCD b % 8 (observed dispatch order; used for speed
    // only).  Each XCD gets a contiguous slab of tiles along the LONGER tile dimension and walks the
    // shorter one fastest, so the operand panel of the long dimension (W rows for the logits GEMM,
    // dlogits columns for dW) is fetched into exactly one XCD's L2 and reused there.  Measured on the
    // logits GEMM (rocprofv3 FETCH_SIZE): 1.12e6 KB fetched per launch with the row-major order vs
    // 52 MB of operands -- every 64-row m-tile re-streamed all of W through the fabric.
    const int TM = (p.M + BM - 1) / BM, TN = (p.N + BN - 1) / BN;
    int mt, nt;
    if (TM * TN < 128 || !p.xcd_order) {       // tiny tile grids (split-K wgrads): plain order
        mt = blockIdx.x / TN; nt = blockIdx.x % TN;
    } else {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int Tl = TM <= TN ? TN : TM, Ts = TM <= TN ? TM : TN;   // long / short tile dimension
        const int qd = Tl >> 3, rd = Tl & 7;                          // balanced slabs: rd XCDs get qd+1
        const int cnt = qd + (xcd < rd ? 1 : 0);
        const int start = xcd * qd + (xcd < rd ? xcd : rd);
        const int il = slot / Ts;
        if (il >= cnt) return;
        const int tl = start + il, ts = slot % Ts;
        if (TM <= TN) { nt = tl; mt = ts; } else { mt = tl; nt = ts; }
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const int batch = blockIdx.z / p.splitk, ks = blockIdx.z % p.splitk;

    // split-K range (multiples of BK)
    const int kt_total = (p.K + BK - 1) / BK;
    const int kt_per = (kt_total + p.splitk - 1) / p.splitk;
    const int kt_begin = ks * kt_per;
    const int kt_end = min(kt_total, kt_begin + kt_per);
    if (kt_begin >= kt_end) return;

    const float* A = p.A + batch * p.sA;
    const float* B = p.B + batch * p.sB;
    float* C = p.C + batch * p.sC;

    // Two register stages of staging data: the interior k-loop keeps the global loads of TWO k-tiles
    // in flight (tile kt+2 is requested while tile kt is multiplied and tile kt+1 moves from its
    // stage to LDS), so a load has two MFMA phases (~1000 cycles) to arrive instead of one.
    // softmax-gradient A operand: per-row lse / label are prefetched with the tile, the transform
    // itself runs in store_tiles (after the MFMAs), so the global loads still overlap compute.
    float4 ra0[NA4], rb0[NB4], ra1[NA4], rb1[NB4];
    float sgl0[NA4], sgl1[NA4];
    long sgy0[NA4], sgy1[NA4];   // labels stay 64-bit here: narrowing at load time would wait for the load
#define T4R_STAGE_PARAMS float4(&ra)[NA4], float4(&rb)[NB4], float(&sgl)[NA4], long(&sgy)[NA4]
#define T4R_S0 ra0, rb0, sgl0, sgy0
#define T4R_S1 ra1, rb1, sgl1, sgy1
    // fp16 operands (PREC 3): the softmax gradient (g/N) (p - onehot) of a large vocabulary lies far below fp16's
    // smallest subnormal (6e-8): p ~ 1e-7 at 10 M items, 1/N ~ 1e-4.  It is scaled into range by a power of two
    // chosen from g/N itself (|g/N| 2^k in [2^13, 2^14)) and the epilogue's alpha undoes it exactly -- the per-launch
    // form of what torch.cuda.amp.GradScaler does for the reference (trainer.py:363-367); it composes with a
    // user-level loss scale, which only changes g.
    // PREC 4: the operands' power-of-two positions (max |.| -> [2^13, 2^14); 1 for an all-zero or non-finite operand)
    float op_sa = 1.f, op_sb = 1.f;
    if constexpr (PREC == 4) {
        // every workgroup reduces the producers' per-workgroup maxima itself (a few hundred floats: L2 hits)
        float ma = 0.f, mb = 0.f;
        for (int i = tid; i < p.n_amax; i += 256) ma = fmaxf(ma, p.amaxA[i]);
        for (int i = tid; i < p.n_amax_b; i += 256) mb = fmaxf(mb, p.amaxB[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ma = fmaxf(ma, __shfl_xor(ma, o, 64)); mb = fmaxf(mb, __shfl_xor(mb, o, 64)); }
        float* red = smem;      // the operand images are not in use yet
        if (lane == 0) { red[2 * wave] = ma; red[2 * wave + 1] = mb; }
        __syncthreads();
        ma = fmaxf(fmaxf(red[0], red[2]), fmaxf(red[4], red[6]));
        mb = fmaxf(fmaxf(red[1], red[3]), fmaxf(red[5], red[7]));
        __syncthreads();
        auto scale_of = [](float a) {
            if (!(a > 0.f) || !(a < 3e38f)) return 1.f;
            int e;
            (void)frexpf(a, &e);
            return ldexpf(1.f, min(14 - e, 100));
        };
        op_sa = scale_of(ma);
        op_sb = scale_of(mb);
    }
    float sg_g = 0.f, sg_unscale = 1.f;
    if (SG) {
        sg_g = (p.sg_gout ? *p.sg_gout : 1.f) / p.sg_rows;
        if (PREC == 3) {
            int ex;
            (void)frexpf(fabsf(sg_g), &ex);
            ex = max(ex, -86);          // a vanishing upstream gradient must not overflow the scale
            sg_unscale = ldexpf(1.f, ex - 14);
            sg_g *= ldexpf(1.f, 14 - ex);
        }
    }

    // staging loads of k-tile kt (all addresses legal, see ld4_clamped)
    auto load_tiles = [&](T4R_STAGE_PARAMS, int kt) __attribute__((always_inline)) {
        const int k0 = kt * BK;
#pragma unroll
        for (int r = 0; r < NA4; ++r) {
            const int idx = tid + r * 256;
            if (TA) {  // A[K][lda], m contiguous: tile row = k, 4 consecutive m
                int k, m4;
                km_map(r, BM / 4, NA4, k, m4);
                const int gk = min(k0 + k, p.K - 1);
                ra[r] = ld4_clamped<VEC>(A + (long)gk * p.lda, m0 + m4, p.M);
                if (SG) { sgl[r] = p.sg_lse[gk]; sgy[r] = p.sg_labels[gk]; }           // rows = k
            } else {   // A[M][lda], k contiguous: tile row = m, 4 consecutive k
                const int m = idx / (BK / 4), k4 = (idx % (BK / 4)) * 4;
                const int gm = min(m0 + m, p.M - 1);
                ra[r] = ld4_clamped<VEC>(A + (long)gm * p.lda, k0 + k4, p.K);
                if (SG) { sgl[r] = p.sg_lse[gm]; sgy[r] = p.sg_labels[gm]; }           // rows = m (L1-resident re-read)
            }
        }
#pragma unroll
        for (int r = 0; r < NB4; ++r) {
            const int idx = tid + r * 256;
            if (!TB) {  // B[K][ldb], n contiguous
                int k, n4;
                km_map(r, BN / 4, NB4, k, n4);
                rb[r] = ld4_clamped<VEC>(B + (long)min(k0 + k, p.K - 1) * p.ldb, n0 + n4, p.N);
            } else {    // B[N][ldb], k contiguous
                const int n = idx / (BK / 4), k4 = (idx % (BK / 4)) * 4;
                rb[r] = ld4_clamped<VEC>(B + (long)min(n0 + n, p.N - 1) * p.ldb, k0 + k4, p.K);
            }
        }
    };

    // softmax-gradient transform of a staged A tile, in registers (SG variants).  In the k-loop it
    // is scheduled UNDER the MFMAs of the resident tile (sched_group_barrier pattern below): the
    // ~50 VALU ops + 4 exp per thread otherwise sit between the MFMA phase and the barrier.
    auto transform_stage = [&](T4R_STAGE_PARAMS, int kt, bool live) __attribute__((always_inline)) {
        const int k0 = kt * BK;
#pragma unroll
        for (int r = 0; r < NA4; ++r) {
            pin4(ra[r]);
            asm volatile("" : "+v"(sgl[r]), "+v"(sgy[r]));
        }
#pragma unroll
        for (int r = 0; r < NA4; ++r) {
            const int idx = tid + r * 256;
            if (TA) {
                int k, m4;
                km_map(r, BM / 4, NA4, k, m4);
                const int gk = k0 + k, gm = m0 + m4;
                ra[r] = softmax_grad4(ra[r], sgl[r], (int)(sgy[r] - p.sg_yoff), gm, sg_g, p);
            } else {
                const int m = idx / (BK / 4), k4 = (idx % (BK / 4)) * 4;
                const int gm = m0 + m, gk = k0 + k4;
                ra[r] = softmax_grad4(ra[r], sgl[r], (int)(sgy[r] - p.sg_yoff), gk, sg_g, p);
            }
        }
    };

    // stage -> LDS image of buffer `buf`.  Only k needs zeroing: rows / columns beyond M / N hold a
    // clamped duplicate of the last row / column and only feed outputs that are never stored, but
    // k >= K (the K tail, the zero tile that pads an odd tile count) would reach valid outputs.
    // The zeroing is a workgroup-uniform branch taken by those tiles only: on interior k-tiles it
    // would cost 18 VALU ops + hazard nops per k-tile between the MFMAs and the barrier.
    auto store_tiles = [&](T4R_STAGE_PARAMS, int buf, int kt, bool live) __attribute__((always_inline)) {
        float* as = As + buf * A_SZ;
        float* bs = Bs + buf * B_SZ;
        const int k0 = kt * BK;
#pragma unroll
        for (int r = 0; r < NA4; ++r) pin4(ra[r]);
#pragma unroll
        for (int r = 0; r < NB4; ++r) pin4(rb[r]);
        if (!live || k0 + BK > p.K) {
#pragma unroll
            for (int r = 0; r < NA4; ++r) {
                const int idx = tid + r * 256;
                int kk, mm;
                km_map(r, BM / 4, NA4, kk, mm);
                if (TA) ra[r] = mask4(ra[r], (live && k0 + kk < p.K) ? 4 : 0);
                else ra[r] = mask4(ra[r], live ? p.K - (k0 + (idx % (BK / 4)) * 4) : 0);
            }
#pragma unroll
            for (int r = 0; r < NB4; ++r) {
                const int idx = tid + r * 256;
                int kk, nn;
                km_map(r, BN / 4, NB4, kk, nn);
                if (!TB) rb[r] = mask4(rb[r], (live && k0 + kk < p.K) ? 4 : 0);
                else rb[r] = mask4(rb[r], live ? p.K - (k0 + (idx % (BK / 4)) * 4) : 0);
            }
        }
        if constexpr (PREC == 4) {
#pragma unroll
            for (int r = 0; r < NA4; ++r) { ra[r].x *= op_sa; ra[r].y *= op_sa; ra[r].z *= op_sa; ra[r].w *= op_sa; }
#pragma unroll
            for (int r = 0; r < NB4; ++r) { rb[r].x *= op_sb; rb[r].y *= op_sb; rb[r].z *= op_sb; rb[r].w *= op_sb; }
        }
        if constexpr (HALF) {
            // fp32 stage -> bf16 / fp16 planes (PREC 1: the exact three-way split)
            uint32_t* ah = reinterpret_cast<uint32_t*>(As) + buf * A_SZ;
            uint32_t* bh = reinterpret_cast<uint32_t*>(Bs) + buf * B_SZ;
            if (A_MK) {
#pragma unroll
                for (int r = 0; r < NA4; ++r) {
                    const int idx = tid + r * 256, m = idx / (BK / 4), k4 = (idx % (BK / 4)) * 4;
                    uint32_t w0[NPL], w1[NPL];
                    cvt_pair<PREC>(ra[r].x, ra[r].y, w0);
                    cvt_pair<PREC>(ra[r].z, ra[r].w, w1);
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        *reinterpret_cast<uint2*>(ah + pl * HA_PLANE + (k4 >> 3) * HA_PITCH + m * 4 + ((k4 >> 2) & 1) * 2) =
                            make_uint2(w0[pl], w1[pl]);
                }
            } else {
#pragma unroll
                for (int rp = 0; rp < NA4 / 2; ++rp) {
                    int k, m4;
                    km_map(2 * rp, BM / 4, NA4, k, m4);
                    uint32_t wx[NPL], wy[NPL], wz[NPL], ww[NPL];
                    cvt_pair<PREC>(ra[2 * rp].x, ra[2 * rp + 1].x, wx);
                    cvt_pair<PREC>(ra[2 * rp].y, ra[2 * rp + 1].y, wy);
                    cvt_pair<PREC>(ra[2 * rp].z, ra[2 * rp + 1].z, wz);
                    cvt_pair<PREC>(ra[2 * rp].w, ra[2 * rp + 1].w, ww);
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        *reinterpret_cast<uint4*>(ah + pl * HA_PLANE + ((k >> 3) * 4 + ((k & 7) >> 1)) * HA_PITCH + m4) =
                            make_uint4(wx[pl], wy[pl], wz[pl], ww[pl]);
                }
            }
            if (B_MK) {
#pragma unroll
                for (int r = 0; r < NB4; ++r) {
                    const int idx = tid + r * 256, n = idx / (BK / 4), k4 = (idx % (BK / 4)) * 4;
                    uint32_t w0[NPL], w1[NPL];
                    cvt_pair<PREC>(rb[r].x, rb[r].y, w0);
                    cvt_pair<PREC>(rb[r].z, rb[r].w, w1);
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        *reinterpret_cast<uint2*>(bh + pl * HB_PLANE + (k4 >> 3) * HB_PITCH + n * 4 + ((k4 >> 2) & 1) * 2) =
                            make_uint2(w0[pl], w1[pl]);
                }
            } else {
#pragma unroll
                for (int rp = 0; rp < NB4 / 2; ++rp) {
                    int k, n4;
                    km_map(2 * rp, BN / 4, NB4, k, n4);
                    uint32_t wx[NPL], wy[NPL], wz[NPL], ww[NPL];
                    cvt_pair<PREC>(rb[2 * rp].x, rb[2 * rp + 1].x, wx);
                    cvt_pair<PREC>(rb[2 * rp].y, rb[2 * rp + 1].y, wy);
                    cvt_pair<PREC>(rb[2 * rp].z, rb[2 * rp + 1].z, wz);
                    cvt_pair<PREC>(rb[2 * rp].w, rb[2 * rp + 1].w, ww);
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        *reinterpret_cast<uint4*>(bh + pl * HB_PLANE + ((k >> 3) * 4 + ((k & 7) >> 1)) * HB_PITCH + n4) =
                            make_uint4(wx[pl], wy[pl], wz[pl], ww[pl]);
                }
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < NA4; ++r) {
            const int idx = tid + r * 256;
            if (TA) {
                const int k = idx / (BM / 4), m4 = (idx % (BM / 4)) * 4;
                *reinterpret_cast<float4*>(as + k * LDA_S + m4) = ra[r];
            } else {
                const int m = idx / (BK / 4), k4 = (idx % (BK / 4)) * 4;
                *reinterpret_cast<float4*>(as + m * LDA_S + swz(m, k4 / 4)) = ra[r];
            }
        }
#pragma unroll
        for (int r = 0; r < NB4; ++r) {
            const int idx = tid + r * 256;
            if (!TB) {
                const int k = idx / (BN / 4), n4 = (idx % (BN / 4)) * 4;
                *reinterpret_cast<float4*>(bs + k * LDB_S + n4) = rb[r];
            } else {
                const int n = idx / (BK / 4), k4 = (idx % (BK / 4)) * 4;
                *reinterpret_cast<float4*>(bs + n * LDB_S + swz(n, k4 / 4)) = rb[r];
            }
        }
    };

    // (a second, independent accumulator chain per tile was measured: no gain -- the 64-cycle
    //  dependent-accumulator latency equals the issue interval of v_mfma_f32_32x32x2_f32)
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int arow = wm * (BM / 2) + (lane & 31);
    const int bcol = wn * (BN / 2) + (lane & 31);
    const int khalf = lane >> 5;

    // MFMA operand fragments of ONE k-tile, held in registers: NH groups of four MFMA steps
    // (physical k = khalf*KH + 4*h + e).  They are fetched from LDS one k-tile ahead (see below).
    // Half path: a group = one K = 16 MFMA step; the lane's fragment of step h is the chunk 2 h + khalf
    // (8 consecutive k = four words) of its row, per plane.
    constexpr int NH = HALF ? BK / 16 : KH / 4;
    constexpr int MFMA_PER_GROUP = HALF ? (PREC == 1 ? 6 : (PREC == 4 ? 3 : 1)) : 4;
    float4 fa[HALF ? 1 : NH][WM], fb[HALF ? 1 : NH][WN];
    uint4 fah[HALF ? NH : 1][WM][NPL], fbh[HALF ? NH : 1][WN][NPL];
    auto read_frag = [&](int buf, int h) __attribute__((always_inline)) {
        if constexpr (HALF) {
            const uint32_t* ah = reinterpret_cast<const uint32_t*>(As) + buf * A_SZ;
            const uint32_t* bh = reinterpret_cast<const uint32_t*>(Bs) + buf * B_SZ;
            const int c = 2 * h + khalf;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    if (A_MK) {
                        fah[h][i][pl] = *reinterpret_cast<const uint4*>(ah + pl * HA_PLANE + c * HA_PITCH + (arow + i * 32) * 4);
                    } else {
                        const uint32_t* q = ah + pl * HA_PLANE + c * 4 * HA_PITCH + arow + i * 32;
                        fah[h][i][pl] = make_uint4(q[0], q[HA_PITCH], q[2 * HA_PITCH], q[3 * HA_PITCH]);
                    }
                }
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    if (B_MK) {
                        fbh[h][j][pl] = *reinterpret_cast<const uint4*>(bh + pl * HB_PLANE + c * HB_PITCH + (bcol + j * 32) * 4);
                    } else {
                        const uint32_t* q = bh + pl * HB_PLANE + c * 4 * HB_PITCH + bcol + j * 32;
                        fbh[h][j][pl] = make_uint4(q[0], q[HB_PITCH], q[2 * HB_PITCH], q[3 * HB_PITCH]);
                    }
                }
            return;
        }
        const float* as = As + buf * A_SZ;
        const float* bs = Bs + buf * B_SZ;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            if (A_MK) {
                fa[h][i] = *reinterpret_cast<const float4*>(as + (arow + i * 32) * LDA_S + swz(arow + i * 32, khalf * (KH / 4) + h));
            } else {
                const float* q = as + (khalf * KH + 4 * h) * LDA_S + arow + i * 32;
                fa[h][i] = make_float4(q[0], q[LDA_S], q[2 * LDA_S], q[3 * LDA_S]);
            }
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            if (B_MK) {
                fb[h][j] = *reinterpret_cast<const float4*>(bs + (bcol + j * 32) * LDB_S + swz(bcol + j * 32, khalf * (KH / 4) + h));
            } else {
                const float* q = bs + (khalf * KH + 4 * h) * LDB_S + bcol + j * 32;
                fb[h][j] = make_float4(q[0], q[LDB_S], q[2 * LDB_S], q[3 * LDB_S]);
            }
        }
    };
    auto mfma_group = [&](int h) __attribute__((always_inline)) {
        if constexpr (HALF) {
            if constexpr (PREC == 1) {
                // six partial products, smallest first; the accumulators of the tile alternate inside a term
#define T4R_TERM(PA, PB)                                                                          \
                _Pragma("unroll") for (int i = 0; i < WM; ++i) _Pragma("unroll") for (int j = 0; j < WN; ++j) \
                    acc[i][j] = mfma_half<PREC>(fah[h][i][PA], fbh[h][j][PB], acc[i][j]);
                T4R_TERM(1, 1) T4R_TERM(2, 0) T4R_TERM(0, 2) T4R_TERM(1, 0) T4R_TERM(0, 1) T4R_TERM(0, 0)
#undef T4R_TERM
            } else if constexpr (PREC == 4) {
#define T4R_TERM(PA, PB)                                                                          \
                _Pragma("unroll") for (int i = 0; i < WM; ++i) _Pragma("unroll") for (int j = 0; j < WN; ++j) \
                    acc[i][j] = mfma_half<PREC>(fah[h][i][PA], fbh[h][j][PB], acc[i][j]);
                T4R_TERM(1, 0) T4R_TERM(0, 1) T4R_TERM(0, 0)
#undef T4R_TERM
            } else {
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j] = mfma_half<PREC>(fah[h][i][0], fbh[h][j][0], acc[i][j]);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][i].x, fb[h][j].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][i].y, fb[h][j].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][i].z, fb[h][j].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h][i].w, fb[h][j].w, acc[i][j], 0, 0, 0);
            }
    };

    // Software pipeline over the workgroup's k-tiles, three stages deep:
    //   tile t   : its fragments are in registers and feed the MFMAs
    //   tile t+1 : moves from its register stage to LDS[buf^1]; after the barrier its first fragment
    //              group is fetched while the LAST MFMA group of tile t still runs, the other groups
    //              right after, so no MFMA waits on an LDS read issued just before it
    //   tile t+2 : requested from memory (two MFMA phases, ~1000 cycles, to arrive)
    // The loop is branch-free and runs over PAIRS of tiles (the two register stages swap roles, so
    // the register roles are identical at every back-edge); an odd tile count is rounded up with an
    // all-zero tile, and requests past the last tile re-read the last tile (legal addresses, data
    // never used).  sched_barriers pin the order: left alone, the scheduler sinks the global loads
    // below the MFMAs to reuse registers and then waits for them right away.
    {
        const int last = kt_end - 1;
        // the first TWO k-tiles are requested together, so the second one's latency runs under the
        // first one's trip through LDS (K = 128 tiles are only 8 k-tiles long: the prologue counts)
        load_tiles(T4R_S0, kt_begin);
        load_tiles(T4R_S1, min(kt_begin + 1, last));
        __builtin_amdgcn_sched_barrier(0);
        if (SG) transform_stage(T4R_S0, kt_begin, true);
        store_tiles(T4R_S0, 0, kt_begin, true);
        __syncthreads();
#pragma unroll
        for (int h = 0; h < NH; ++h) read_frag(0, h);
#define T4R_STEP(SLOAD, SSTORE, BUFN, TNEXT)                          \
        load_tiles(SLOAD, min((TNEXT) + 1, last));                      \
        __builtin_amdgcn_sched_barrier(0);                              \
        if (SG) transform_stage(SSTORE, TNEXT, (TNEXT) < kt_end);       \
        _Pragma("unroll") for (int h = 0; h < NH - 1; ++h) mfma_group(h); \
        if (SG) {   /* one MFMA, then a slice of the transform's VALU work, ... */ \
            _Pragma("unroll") for (int g = 0; g < MFMA_PER_GROUP * (NH - 1) * WM * WN; ++g) { \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      \
                __builtin_amdgcn_sched_group_barrier(0x402, 16, 0);     \
            }                                                           \
        }                                                               \
        __builtin_amdgcn_sched_barrier(0);                              \
        store_tiles(SSTORE, BUFN, TNEXT, (TNEXT) < kt_end);             \
        __syncthreads();                                                \
        read_frag(BUFN, 0);                                             \
        __builtin_amdgcn_sched_barrier(0);                              \
        mfma_group(NH - 1);                                             \
        __builtin_amdgcn_sched_barrier(0);                              \
        _Pragma("unroll") for (int h = 1; h < NH; ++h) read_frag(BUFN, h);
        for (int kt = kt_begin; kt < kt_end; kt += 2) {
            T4R_STEP(T4R_S0, T4R_S1, 1, kt + 1)
            T4R_STEP(T4R_S1, T4R_S0, 0, kt + 2)
        }
#undef T4R_STEP
    }

    // epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // The mode is workgroup-uniform: it is decided ONCE and each mode has its own straight-line
    // store loop (the per-element switch cost ~30 scalar/vector instructions per output element,
    // a quarter of the MFMA time of a K = 128 tile).
    const float alpha = PREC == 4 ? (p.alpha / op_sa) / op_sb : p.alpha * sg_unscale;
    if constexpr (RANK) {
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    const int rr = min(row, p.M - 1);
                    const float v = alpha * acc[i][j][r], t = p.rk_thr[rr];
                    const bool beats = col < p.N && (v > t || (v == t && col < (int)p.rk_label[rr]));
                    const unsigned long long m = __ballot(beats);
                    const int cnt = __popc((unsigned)(khalf ? (m >> 32) : (m & 0xffffffffull)));
                    if ((lane & 31) == 0 && row < p.M && cnt) atomicAdd(p.rk_count + row, cnt);
                }
            }
        }
        return;
    }
    const bool rows_full = m0 + BM <= p.M;
    auto for_each_out = [&](auto fn) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int col = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
                if (col >= p.N) continue;
                const int row0 = m0 + wm * (BM / 2) + i * 32 + 4 * khalf;
                float* c0 = C + (long)row0 * p.ldc + col;
                float dm[16];
                if (EDROP) {      // epilogue dropout masks of the fragment (N % 4 == 0: whole quads are in range together)
                    if ((p.N & 3) == 0) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float m4[4];
                            drop_scale_quad(p.drop, (unsigned long long)(row0 + 8 * g), (unsigned long long)p.N, col, m4);
                            dm[4 * g] = m4[0]; dm[4 * g + 1] = m4[1]; dm[4 * g + 2] = m4[2]; dm[4 * g + 3] = m4[3];
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            dm[r] = drop_scale(p.drop, (unsigned long long)(row0 + (r & 3) + 8 * (r >> 2)) * p.N + col);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (!rows_full && row0 + dr >= p.M) continue;
                    fn(c0 + (long)dr * p.ldc, row0 + dr, col, alpha * acc[i][j][r], EDROP ? dm[r] : 1.f);
                }
            }
        }
    };
    if (p.splitk > 1) {
        if (p.part) {
            float* P = p.part + (long)blockIdx.z * ((long)p.M * p.ldc);
            for_each_out([&](float* cp, int, int, float v, float) __attribute__((always_inline)) { P[cp - C] = v; });
        } else {
            for_each_out([&](float* cp, int, int, float v, float) __attribute__((always_inline)) { atomicAdd(cp, v); });
        }
    } else if (p.epilogue == EPI_NONE) {
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
                // C = dropout(x + bias) + residual   (GPT-2: hidden + resid_dropout(c_proj(...)))
                v += bv;
                if (EDROP) v *= dmask;
                v += p.aux[(long)row * p.ldaux + col];
            }
            if (acc_c) v += *cp;
            *cp = v;
        });
    }
}

template <int BM, int BN, int BK, bool TA, bool TB, int FEAT, bool VEC, int PREC = 0>
static int launch_vec(const GemmParams& p, int batch, hipStream_t stream) {
    static long pad = -1;   // experiment knob: extra LDS per workgroup = fewer resident workgroups per CU
    if (pad < 0) { const char* e = t4r_exp_getenv("T4R_GEMM_LDS_PAD"); pad = e ? atol(e) : 0; }
    const size_t smem = gemm_lds_bytes<BM, BN, BK, TA, TB, PREC>() + (size_t)pad;
    static T4rLdsAttr attr_set;
    t4r_ensure_dynamic_lds((const void*)gemm_f32_kernel<BM, BN, BK, TA, TB, FEAT, VEC, PREC>, smem, attr_set);
    const int TM = (p.M + BM - 1) / BM, TN = (p.N + BN - 1) / BN;
    const int Tl = TM <= TN ? TN : TM, Ts = TM <= TN ? TM : TN;
    const int gx = (TM * TN < 128 || !p.xcd_order) ? TM * TN : 8 * ((Tl + 7) / 8) * Ts;   // must match the kernel's decode
    dim3 grid(gx, 1, batch * p.splitk);
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, TA, TB, FEAT, VEC, PREC>), grid, dim3(256), smem, stream, p);
    T4R_LAUNCH_CHECK();
    return 0;
}

