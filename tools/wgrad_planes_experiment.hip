// EXPERIMENT, not part of the library (round 3): weight gradients from pre-cut bf16 planes with gfx950's transposing LDS read.
// Kept for the record of what was measured; compile against csrc/ (-I transformers4rec_amd/csrc) to revive it.
// Result at BASELINE configs[1] (d W1 = d ffpre^T @ h1, T = 20 480, 512 x 128 output, both operands as planes in HBM):
//   this kernel, fp32 atomics into C ............ 33-43 us   (same-address atomics of 64 splits serialise: 24 us with plain stores)
//   general split-K GEMM (gemm_kernel.h, PREC 1)  31-34 us stand-alone, 32 us in the step
// 1.24 us per 32-token k-step against 0.64 us of MFMA issue (one wave per SIMD, 108 KB of LDS per workgroup); with the extra
// 21 MB per operand the producers would have to write as planes (6 bytes per element instead of 4) there is nothing to
// gain over the general kernel, so the layer keeps the general kernel (now with the two-stage deterministic split-K,
// gemm_f32.hip).  What IS reusable: the measured semantics of ds_read_b64_tr_b16 below and the conflict-free pitch rule.
//
// Weight gradients of the XLNet layer from PRE-CUT operands:  C[M, N] += A^T @ B with the token index as the contraction
// index, A [T, M] and B [T, N] given as the three bf16 planes (x = hi + mid + lo exactly) that the token-tile kernels of
// xlnet_fused.hip / xlnet_fused_attn.hip already hold in LDS when they produce these rows.
//
// Replaces, for the feed-forward weights, the split-K launches of the general GEMM (gemm_kernel.h, PREC 1) in
// t4r_xlnet_layer_bwd:  d W1 = d ffpre^T @ h1,  d W2 = d ffout^T @ ffact  (HF modeling_xlnet.py:478-486 ff / its autograd, via
// transformers4rec/torch/block/transformer.py:179-199).  The general kernel re-cuts both fp32 operand tiles every time a
// workgroup stages them -- with 64 x 64 tiles the VALU work of the cuts equals the matrix-core time, and VALU instructions
// do not issue under MFMAs of the same SIMD (tools/mfma_valu_overlap.hip) -- 40 us per FF weight gradient at BASELINE
// configs[1] against ~14 us of HBM time.  Here nothing is converted: plane tiles go global -> LDS as they are and the MFMA
// fragments (eight consecutive TOKENS of one feature: the k index runs down the rows of the row-major planes) come out of
// LDS through gfx950's transposing read ds_read_b64_tr_b16.
//
// ds_read_b64_tr_b16, as measured (tools/ds_read_tr_probe.hip): inside each group of 16 lanes, lane i supplies the address of 4
// consecutive bf16; output lane c receives, for j = 0..3, element (c % 4) of what lane 4 j + c / 4 addressed.  With lane
// i pointing at row (i >> 2), columns 4 (i & 3) .. + 3 of a row-major block, lane c ends up with column c of rows 0..3.
// Two reads (rows 4 g .. 4 g + 3 and 16 + 4 g .. 16 + 4 g + 3 for lane group g = lane >> 4) give the eight k-slots of one
// v_mfma_f32_16x16x32_bf16 operand; A and B use the same slot -> token map, so the contraction is over all 32 tokens of
// a k-step.  LDS rows are pitched at (width + 16) bf16: the eight rows read by 32 lanes then start 32 bytes apart modulo
// 256 and every read is bank-conflict free.
//
// Work split: tile = 128 columns of the WIDE operand x all D columns of the narrow one; the token range is cut into
// `splits` pieces and the partial tiles are added with fp32 atomics (as the general split-K path does).  The m-tiles of
// one token range sit on one XCD (workgroup b runs on XCD b % 8), so the narrow operand's rows are fetched into that L2
// once.
#include "xlnet_fused.h"

namespace {

typedef short v4s __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32x2 ds_tr16(const uint16_t* p) {
    const v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p));
    return __builtin_bit_cast(u32x2, r);
}

struct WgParams {
    const uint16_t* wide; long wide_pl; int ld_wide; int n_mt;      // [3][T][ld_wide] planes, n_mt = width / 128
    const uint16_t* narrow; long narrow_pl; int ld_narrow;          // [3][T][ld_narrow] planes, D columns used
    float* C; int ldc; int T; int splits; int ks_per;
};

// SWAP = false: C[wide col][narrow col];  SWAP = true: C[narrow col][wide col]
template <int D, bool SWAP>
__global__ __launch_bounds__(256) void wgrad_planes_kernel(WgParams p) {
    constexpr int PA = 128 + 16, PB = D + 16;            // LDS row pitches (bf16 elements)
    constexpr int A_PL = 32 * PA, B_PL = 32 * PB;
    constexpr int BUF = 3 * (A_PL + B_PL);
    constexpr int NA = 6;                                // 16-byte chunks per thread per k-step, wide operand
    constexpr int B_CH = 3 * 32 * D / 8, NB = (B_CH + 255) / 256;
    constexpr int NBW = D / 32;                          // 16-column blocks of the narrow operand per wave
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int mt = slot % p.n_mt, split = (slot / p.n_mt) * 8 + xcd;
    if (split >= p.splits) return;
    const int nks = (p.T + 31) / 32;
    const int ks0 = split * p.ks_per, ks1 = min(nks, ks0 + p.ks_per);
    if (ks0 >= ks1) return;

    // staging registers: THREE k-steps of plane tiles in flight per workgroup (one wave per SIMD and one workgroup per CU
    // -- 108 KB of LDS -- leave nothing else to cover the ~2 us of HBM latency: with one tile in flight a k-step took as
    // long as a load, 31-41 us per launch instead of the ~14 us the bytes need)
    struct Stage { u32x4 a[NA], b[NB]; };
    auto g_load = [&](Stage& s, int ks) __attribute__((always_inline)) {
        // every load is unconditional (a load under a branch makes the compiler wait for ALL outstanding loads at the next
        // use: s_waitcnt vmcnt(0) in front of every barrier).  Steps past the end of the range (ks >= ks1, never multiplied)
        // fetch one and the same 16 bytes.
        const int k0 = ks * 32;
        const bool live = ks < ks1;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = tid + 256 * i, pl = c >> 9, row = (c & 511) >> 4, ch = c & 15;
            const int gr = min(k0 + row, p.T - 1);
            const uint16_t* src = p.wide + pl * p.wide_pl + (long)gr * p.ld_wide + mt * 128 + ch * 8;
            s.a[i] = ldq(live ? src : p.wide);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int c = min(tid + 256 * i, B_CH - 1), pl = c / (32 * D / 8), rem = c % (32 * D / 8);
            const int row = rem / (D / 8), ch = rem % (D / 8);
            const int gr = min(k0 + row, p.T - 1);
            const uint16_t* src = p.narrow + pl * p.narrow_pl + (long)gr * p.ld_narrow + ch * 8;
            s.b[i] = ldq(live ? src : p.narrow);
        }
    };
    // rows past T (the tail k-step only) are zeroed in ONE operand: their products vanish
    auto s_store = [&](const Stage& s, int buf, int ks) __attribute__((always_inline)) {
        uint16_t* base = smem + buf * BUF;
        const bool tail = ks * 32 + 32 > p.T;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = tid + 256 * i, pl = c >> 9, row = (c & 511) >> 4, ch = c & 15;
            u32x4 v = s.a[i];
            if (tail && ks * 32 + row >= p.T) v = u32x4{0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(base + pl * A_PL + row * PA + ch * 8) = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int c = tid + 256 * i;
            if (B_CH % 256 == 0 || c < B_CH) {
                const int pl = c / (32 * D / 8), rem = c % (32 * D / 8), row = rem / (D / 8), ch = rem % (D / 8);
                *reinterpret_cast<u32x4*>(base + 3 * A_PL + pl * B_PL + row * PB + ch * 8) = s.b[i];
            }
        }
    };

    f32x4 acc[4][NBW];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NBW; ++b) acc[a][b] = zero4();
    // this lane's source position inside a [4 rows][16 columns] block of its lane group's rows
    const int li = lane & 15, lg = lane >> 4;
    const int a_off = (4 * lg + (li >> 2)) * PA + wm * 64 + 4 * (li & 3);
    const int b_off = (4 * lg + (li >> 2)) * PB + wn * (D / 2) + 4 * (li & 3);

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const uint16_t* ab = smem + buf * BUF + a_off;
        const uint16_t* bb = smem + buf * BUF + 3 * A_PL + b_off;
        u32x4 bf[NBW][3];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const u32x2 lo = ds_tr16(bb + pl * B_PL + nb * 16), hi = ds_tr16(bb + pl * B_PL + nb * 16 + 16 * PB);
                bf[nb][pl] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            u32x4 af[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const u32x2 lo = ds_tr16(ab + pl * A_PL + mb * 16), hi = ds_tr16(ab + pl * A_PL + mb * 16 + 16 * PA);
                af[pl] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
#define T4R_PROD(PX, PY)                                                                          \
            _Pragma("unroll") for (int nb = 0; nb < NBW; ++nb)                                    \
                acc[mb][nb] = SWAP ? mfma_bf(bf[nb][PY], af[PX], acc[mb][nb]) : mfma_bf(af[PX], bf[nb][PY], acc[mb][nb]);
            T4R_SIX(T4R_PROD)
#undef T4R_PROD
        }
    };
    // step `it` multiplies tile ks0 + it (LDS buffer it & 1); its staging registers `cur` were stored one step earlier and
    // take tile + 3 now; tile + 1 (registers `nxt`, requested two steps ago) moves to the other LDS buffer behind the MFMAs
    const int n = ks1 - ks0;
    auto step = [&](int it, Stage& cur, const Stage& nxt) __attribute__((always_inline)) {
        g_load(cur, ks0 + it + 3);
        compute(it & 1);
        s_store(nxt, (it + 1) & 1, ks0 + it + 1);       // past the end: a tile nobody reads
        __syncthreads();
    };
    Stage s0, s1, s2;
    g_load(s0, ks0);
    g_load(s1, ks0 + 1);
    g_load(s2, ks0 + 2);
    s_store(s0, 0, ks0);
    __syncthreads();
    int it = 0;
    for (; it + 3 <= n; it += 3) {
        step(it, s0, s1);
        step(it + 1, s1, s2);
        step(it + 2, s2, s0);
    }
    if (it < n) {
        step(it, s0, s1);
        if (it + 1 < n) step(it + 1, s1, s2);
    }
    // accumulator lane: rows 4 lg + r, column li of the 16 x 16 block
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const int m0 = mt * 128 + wm * 64 + mb * 16, n0 = wn * (D / 2) + nb * 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* dst = SWAP ? p.C + (long)(n0 + 4 * lg + r) * p.ldc + m0 + li
                                  : p.C + (long)(m0 + 4 * lg + r) * p.ldc + n0 + li;
#ifdef T4R_WG_NOATOMIC
                *dst = acc[mb][nb][r];
#else
                atomicAdd(dst, acc[mb][nb][r]);
#endif
            }
        }
}

// fp32 [rows, cols] (pitch ld) -> planes [3][rows][cols]: the stand-alone form of the cuts the fused kernels do in their
// epilogues (tests, tools, and operands no fused kernel produces)
__global__ __launch_bounds__(256) void cut_planes_kernel(const float* __restrict__ src, long ld, long rows, int cols,
                                                          uint16_t* __restrict__ dst) {
    const long n4 = rows * (cols / 4);
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const long r = i / (cols / 4);
    const int c = (int)(i % (cols / 4)) * 4;
    const float4 v = ld4(src + r * ld + c);
    uint32_t w0[3], w1[3];
    cut3(v.x, v.y, w0);
    cut3(v.z, v.w, w1);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
        *reinterpret_cast<uint2*>(dst + (long)pl * rows * cols + r * cols + c) = make_uint2(w0[pl], w1[pl]);
}

int wg_splits(int n_mt, int nks) {
    static int target = -1;
    if (target < 0) { const char* e = getenv("T4R_WGRAD_WGS"); target = e ? atoi(e) : 256; }
    return max(1, min(nks, target / n_mt));
}

}  // namespace

extern "C" int t4r_cut_planes(void* stream, const float* src, long ld, long rows, int cols, void* dst) {
    if (rows <= 0 || cols <= 0) return 0;
    T4R_CHECK_ARG(src && dst && cols % 4 == 0 && ld % 4 == 0, "cut_planes: null pointer or width / pitch not a multiple of 4");
    const long n4 = rows * (cols / 4);
    hipLaunchKernelGGL(cut_planes_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld, rows,
                       cols, (uint16_t*)dst);
    T4R_LAUNCH_CHECK();
    return 0;
}

extern "C" int t4r_wgrad_planes_supported(int wide, int D) {
    return wide > 0 && wide % 128 == 0 && (D == 32 || D == 64 || D == 128);
}

// C += wide^T @ narrow (transposed_out = 0: C [wide, D]) or narrow^T @ wide (transposed_out = 1: C [D, wide]).
// wide / narrow: plane 0 of [3][T][ld] bf16 planes (plane stride in elements); C is ACCUMULATED into (fp32 atomics).
extern "C" int t4r_wgrad_planes(void* stream, const void* wide, long wide_plane, int ld_wide, int width, const void* narrow,
                                long narrow_plane, int ld_narrow, int D, int T, float* C, int ldc, int transposed_out) {
    if (T <= 0) return 0;
    T4R_CHECK_ARG(t4r_wgrad_planes_supported(width, D) && wide && narrow && C, "wgrad_planes: unsupported widths or null pointer");
    T4R_CHECK_ARG(ld_wide % 8 == 0 && ld_narrow % 8 == 0 && wide_plane % 8 == 0 && narrow_plane % 8 == 0 &&
                  ((uintptr_t)wide & 15) == 0 && ((uintptr_t)narrow & 15) == 0, "wgrad_planes: planes must be 16-byte aligned with pitches multiple of 8");
    WgParams p;
    p.wide = (const uint16_t*)wide; p.wide_pl = wide_plane; p.ld_wide = ld_wide; p.n_mt = width / 128;
    p.narrow = (const uint16_t*)narrow; p.narrow_pl = narrow_plane; p.ld_narrow = ld_narrow;
    p.C = C; p.ldc = ldc; p.T = T;
    const int nks = (T + 31) / 32;
    p.splits = wg_splits(p.n_mt, nks);
    p.ks_per = (nks + p.splits - 1) / p.splits;
    p.splits = (nks + p.ks_per - 1) / p.ks_per;
    const dim3 grid(p.n_mt * 8 * ((p.splits + 7) / 8));
    hipStream_t st = (hipStream_t)stream;
#define T4R_WG_CASE(DD, SW)                                                                                         \
    {                                                                                                               \
        constexpr size_t smem = (size_t)2 * 3 * 32 * ((128 + 16) + (DD + 16)) * 2;                                   \
        static bool once = false;                                                                                   \
        if (!once) {                                                                                                \
            (void)hipFuncSetAttribute((const void*)wgrad_planes_kernel<DD, SW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
            once = true;                                                                                            \
        }                                                                                                           \
        hipLaunchKernelGGL((wgrad_planes_kernel<DD, SW>), grid, dim3(256), smem, st, p);                            \
    }
    if (transposed_out) {
        if (D == 128) T4R_WG_CASE(128, true) else if (D == 64) T4R_WG_CASE(64, true) else T4R_WG_CASE(32, true)
    } else {
        if (D == 128) T4R_WG_CASE(128, false) else if (D == 64) T4R_WG_CASE(64, false) else T4R_WG_CASE(32, false)
    }
#undef T4R_WG_CASE
    T4R_LAUNCH_CHECK();
    return 0;
}
