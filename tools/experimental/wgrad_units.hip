// EXPERIMENT (round 6, compiled only into the A/B variant library): weight gradients of the transformer body as READ-ONCE,
// CUT-ONCE 128 x 128 output units in the two-way fp16 form.
//
//   C[m][n] += sum_t A[t][m] B[t][n],  t = 20 480 token rows (40 960 for d r), C = 128 x 128 ... 512 x 128
// (torch.nn.Linear's weight gradient x^T dy of the q / k / v / o / r projections and the two feed-forward matrices of an XLNet
// layer: HF modeling_xlnet.py :251-258, :266, :142-152, XLNetFeedForward :296-303, as called by
// transformers4rec/torch/block/transformer.py:179-199; autograd of those products).
//
// The general kernel (gemm_kernel.h, PREC 4) walks such a product as 64 x 64 output tiles: an operand row block is re-read and
// re-cut (fp32 -> two fp16 planes) N / 64 resp. M / 64 times.  Here a workgroup owns a 128 x 128 output unit for its range of token
// rows: per k-step of 32 rows it reads 32 x 128 floats of each operand ONCE, cuts them ONCE into row-major hi | lo planes in LDS
// (global power-of-two positions from the producers' maxima, as PREC 4), and feeds v_mfma_f32_16x16x32_f16 through gfx950's
// transposing LDS read (ds_read_b64_tr_b16: the k index runs down the rows of the planes; layout and conflict-free pitch as
// measured in tools/wgrad_planes_experiment.hip).  Partial units go to the split-K sink of the layer backward (fixed-order
// reduction: bit-reproducible, as before).
//
//   workgroup   256 threads = 4 waves as 2 (m) x 2 (n): a wave owns 64 x 64 = 4 x 4 blocks of 16 x 16, 16 accumulators
//   k-step      32 token rows: 8 float4 per thread in flight under the previous step's products, cut behind them
//   LDS         2 buffers x (A hi | A lo | B hi | B lo) x 32 rows x (128 + 16) halves = 73.7 KB: two workgroups per CU
#include "gemm_kernel.h"
#include "xlnet_fused.h"

namespace {

typedef short v4s __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 ds_tr16(const uint16_t* p) {
    const v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(p));
    return __builtin_bit_cast(u32x2, r);
}

constexpr int WU_P = 128 + 16;            // plane pitch in halves
constexpr int WU_PL = 32 * WU_P;          // one plane of a k-step
constexpr int WU_BUF = 4 * WU_PL;         // A hi | A lo | B hi | B lo

struct WgradUnits {
    const float *A, *B;
    float* part;
    long lda, ldb, ldc, sA, sB;
    int M, K, kper, splits, mb, nb;
    float alpha;
    const float *amaxA, *amaxB;
    int n_amax, n_amax_b;
};

__global__ __launch_bounds__(256, 2) void wgrad_units_kernel(WgradUnits p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int bid = blockIdx.x;
    const int in = bid % p.nb; bid /= p.nb;
    const int im = bid % p.mb; bid /= p.mb;
    const int s = bid % p.splits, b = bid / p.splits;
    const float* A = p.A + b * p.sA + im * 128;
    const float* B = p.B + b * p.sB + in * 128;
    const int k0 = s * p.kper, k1 = min(p.K, k0 + p.kper);

    // the operands' power-of-two positions (max |.| -> [2^13, 2^14)), from the producers' per-workgroup maxima
    float sa, sb;
    {
        float ma = 0.f, mb = 0.f;
        for (int i = tid; i < p.n_amax; i += 256) ma = fmaxf(ma, p.amaxA[i]);
        for (int i = tid; i < p.n_amax_b; i += 256) mb = fmaxf(mb, p.amaxB[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ma = fmaxf(ma, __shfl_xor(ma, o, 64)); mb = fmaxf(mb, __shfl_xor(mb, o, 64)); }
        float* red = reinterpret_cast<float*>(smem);
        if (lane == 0) { red[2 * wave] = ma; red[2 * wave + 1] = mb; }
        __syncthreads();
        ma = fmaxf(fmaxf(red[0], red[2]), fmaxf(red[4], red[6]));
        mb = fmaxf(fmaxf(red[1], red[3]), fmaxf(red[5], red[7]));
        __syncthreads();
        sa = pow2_scale(ma);
        sb = pow2_scale(mb);
    }

    // this thread's four rows of a k-step (row = tid / 32 + 8 i), 16 bytes at column 4 (tid % 32) of both operands
    const int srow = tid >> 5, c4 = (tid & 31) * 4;
    float4 ra[4], rb[4];
    auto request = [&](int k) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = min(k + srow + 8 * i, p.K - 1);
            ra[i] = ld4(A + row * p.lda + c4);
            rb[i] = ld4(B + row * p.ldb + c4);
        }
    };
    auto deposit = [&](int k, int buf) __attribute__((always_inline)) {
        uint16_t* base = smem + buf * WU_BUF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 8 * i;
            const bool live = k + row < k1;                 // rows past the split's range contribute zeros (one operand is enough)
            uint32_t a0[2], a1[2], b0[2], b1[2];
            cut2h(live ? ra[i].x * sa : 0.f, live ? ra[i].y * sa : 0.f, a0);
            cut2h(live ? ra[i].z * sa : 0.f, live ? ra[i].w * sa : 0.f, a1);
            cut2h(rb[i].x * sb, rb[i].y * sb, b0);
            cut2h(rb[i].z * sb, rb[i].w * sb, b1);
            uint16_t* d = base + row * WU_P + c4;
            *reinterpret_cast<uint2*>(d) = make_uint2(a0[0], a1[0]);
            *reinterpret_cast<uint2*>(d + WU_PL) = make_uint2(a0[1], a1[1]);
            *reinterpret_cast<uint2*>(d + 2 * WU_PL) = make_uint2(b0[0], b1[0]);
            *reinterpret_cast<uint2*>(d + 3 * WU_PL) = make_uint2(b0[1], b1[1]);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = zero4();
    // this lane's source position inside a [4 rows][16 columns] block of its lane group's rows (ds_read_b64_tr_b16: lane c of a
    // group of 16 receives column c of the four rows the group addressed)
    const int li = lane & 15, lg = lane >> 4;
    const int a_off = (4 * lg + (li >> 2)) * WU_P + wm * 64 + 4 * (li & 3);
    const int b_off = (4 * lg + (li >> 2)) * WU_P + wn * 64 + 4 * (li & 3);

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const uint16_t* ab = smem + buf * WU_BUF + a_off;
        const uint16_t* bb = smem + buf * WU_BUF + 2 * WU_PL + b_off;
        u32x4 bf[4][2];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const u32x2 lo = ds_tr16(bb + pl * WU_PL + nb * 16), hi = ds_tr16(bb + pl * WU_PL + nb * 16 + 16 * WU_P);
                bf[nb][pl] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            u32x4 af[2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const u32x2 lo = ds_tr16(ab + pl * WU_PL + mb * 16), hi = ds_tr16(ab + pl * WU_PL + mb * 16 + 16 * WU_P);
                af[pl] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = mfma_h(af[1], bf[nb][0], acc[mb][nb]);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = mfma_h(af[0], bf[nb][1], acc[mb][nb]);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = mfma_h(af[0], bf[nb][0], acc[mb][nb]);
        }
    };

    request(k0);
    deposit(k0, 0);
    __syncthreads();
    int buf = 0;
    for (int k = k0; k < k1; k += 32) {
        const bool more = k + 32 < k1;
        if (more) request(k + 32);              // in flight under this step's products
        compute(buf);
        if (more) deposit(k + 32, buf ^ 1);     // the other buffer: every wave left it at the previous barrier
        __syncthreads();
        buf ^= 1;
    }
    // accumulator lane: rows 4 lg + r, column li of the 16 x 16 block
    const float un = (p.alpha / sa) / sb;
    float* out = p.part + ((long)(b * p.splits + s) * p.M + im * 128 + wm * 64) * p.ldc + in * 128 + wn * 64;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(long)(mb * 16 + 4 * lg + r) * p.ldc + nb * 16 + li] = un * acc[mb][nb][r];
}

}  // namespace

// token rows per split: a multiple of 32; the layer's split-K sink is sized for K / 320 + 2 splits per product
void t4r_wgrad_units_plan(int K, int* splits, int* kper) {
    static const int per = [] { const char* e = getenv("T4R_WGRAD_UNITS_ROWS"); return e ? atoi(e) : 640; }();
    int kp = (per + 31) / 32 * 32;
    if (kp < 320) kp = 320;
    *kper = kp;
    *splits = (K + kp - 1) / kp;
}

bool t4r_wgrad_units_ok(const GemmParams& p) {
    static const int on = [] { const char* e = getenv("T4R_WGRAD_UNITS"); return e ? atoi(e) : 0; }();
    return on && p.M % 128 == 0 && p.N % 128 == 0 && p.K >= 1280 && p.vecA && p.vecB && p.accumulate && p.epilogue == EPI_NONE &&
           !p.sg_lse && !p.rk_thr && p.ldc == p.N && p.amaxA && p.amaxB;
}

int t4r_wgrad_units_launch(const GemmParams& p, int batch, int kper, float* part, hipStream_t st) {
    WgradUnits q;
    q.A = p.A; q.B = p.B; q.part = part; q.lda = p.lda; q.ldb = p.ldb; q.ldc = p.ldc; q.sA = p.sA; q.sB = p.sB;
    q.M = p.M; q.K = p.K; q.kper = kper; q.splits = p.splitk; q.mb = p.M / 128; q.nb = p.N / 128; q.alpha = p.alpha;
    q.amaxA = p.amaxA; q.amaxB = p.amaxB; q.n_amax = p.n_amax; q.n_amax_b = p.n_amax_b;
    const size_t smem = (size_t)2 * WU_BUF * sizeof(uint16_t);
    static T4rLdsAttr once;
    t4r_ensure_dynamic_lds((const void*)wgrad_units_kernel, smem, once);
    const long grid = (long)batch * q.splits * q.mb * q.nb;
    hipLaunchKernelGGL(wgrad_units_kernel, dim3((unsigned)grid), dim3(256), smem, st, q);
    T4R_LAUNCH_CHECK();
    return 0;
}
