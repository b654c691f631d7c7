// Weight gradients of the transformer body as a K-STREAMING product (round 4):  C[m][n] += sum_t A[t][m] B[t][n] with
// t = 20 480 token rows and a 128 x 128 ... 128 x 512 output -- torch.nn.Linear's weight gradient x^T dy for the q / k / v / o / r
// projections and the two feed-forward matrices of an XLNet layer (HF modeling_xlnet.py :251-258, :266, :142-152, XLNetFeedForward
// :296-303, as called by transformers4rec/torch/block/transformer.py:179-199; autograd of those products).
//
// The general kernel (gemm_kernel.h) walks such a product as 64 x 64 output tiles x 64 splits of 20 k-tiles: 256 small workgroups
// per 128 x 128 output whose time is a chain of 20 dependent load -> LDS -> matrix steps (20 us alone, 48-60 us inside the step,
// where they share the CUs with the critical chain), both operands re-read per tile.  Here a workgroup owns a WHOLE 128 x 128
// output block for its range of token rows: the rows of both operands are read once, 64 at a time, straight into LDS (every
// request of a batch in flight at once, the next batch requested before this one's products), and all 128 x 128 x 64
// multiply-adds of a batch issue from 8 waves on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 sums: the arithmetic of the
// fp32 kernel it replaces; the feed-forward pair drops its two-way fp16 form and the operand maxima with it).
//
//   wave w          output rows 16 w .. 16 w + 15 of the block, all 128 columns: 8 accumulators
//   k-step          4 token rows: A value = As[4 s + g][16 w + n] (one 32-bit LDS read), B values = Bs[4 s + g][4 n .. 4 n + 3] and
//                   [64 + 4 n ..] (two 128-bit reads) feed 8 instructions -- accumulator j holds output columns 4 n + j: a lane
//                   owns 4 CONSECUTIVE columns of a row and stores 16 bytes
// MEASURED (round 4, BASELINE configs[1], profiles/r04_ws_kernel_stats.csv) AND NOT ENABLED BY DEFAULT (T4R_WGRAD_STREAM=1 turns
// it on): 105.8 us per launch inside the step (46 at best) against 48-60 us for the launches it replaces, and the step went from
// 2.93 to 3.43 ms.  The arithmetic above misses what matters in the step: these products run on the side streams UNDER the critical
// chain, and a 512-thread, 74 KB-LDS workgroup that keeps the matrix pipe busy (35 GFLOP at the fp32 rate = 223 us of the whole
// chip per step) takes the CUs and the pipe away from that chain (attention backward 72 -> 115 us, d h 45 -> 76 us), while the
// small, latency-bound workgroups of the general kernel slip in beside it.  Kept as the measured counter-example; the default
// path is unchanged.
//
//   partial sums    split s stores its block into the split-K sink's image (gemm_f32.hip); the fixed-order reduction launch of the
//                   layer adds them: bit-reproducible, as before
#include "gemm_kernel.h"
#include "xlnet_fused.h"

namespace {

constexpr int WS_ROWS = 64;       // token rows per batch
constexpr int WS_P = 144;         // LDS pitch in floats: 144 mod 64 = 16 -> the four row groups of a k-step read four different bank quarters

struct WgradStream {
    const float* A; const float* B; float* part;
    long lda, ldb, ldc, sA, sB;
    int M, K, kper, splits, mb, nb;
    float alpha;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__global__ __launch_bounds__(512) void wgrad_stream_kernel(WgradStream p) {
    extern __shared__ float smem[];
    float* As = smem;
    float* Bs = smem + WS_ROWS * WS_P;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, g = lane >> 4;
    int bid = blockIdx.x;
    const int in = bid % p.nb; bid /= p.nb;
    const int im = bid % p.mb; bid /= p.mb;
    const int s = bid % p.splits, b = bid / p.splits;
    const float* A = p.A + b * p.sA + im * 128;
    const float* B = p.B + b * p.sB + in * 128;
    const int k0 = s * p.kper, k1 = min(p.K, k0 + p.kper);
    // this thread's four rows of a batch (row = tid / 32 + 16 i), 16 bytes at column 4 (tid % 32) of both operands
    const int srow = tid >> 5, c4 = (tid & 31) * 4;
    float4 ra[4], rb[4];
    auto request = [&](int k) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = min(k + srow + 16 * i, p.K - 1);
            ra[i] = ld4(A + row * p.lda + c4);
            rb[i] = ld4(B + row * p.ldb + c4);
        }
    };
    auto deposit = [&](int k) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool live = k + srow + 16 * i < k1;       // rows past the split's range contribute zeros
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(As + (srow + 16 * i) * WS_P + c4) = live ? ra[i] : z;
            *reinterpret_cast<float4*>(Bs + (srow + 16 * i) * WS_P + c4) = live ? rb[i] : z;
        }
    };
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = zero4();
    request(k0);
    deposit(k0);
    __syncthreads();
    for (int k = k0; k < k1; k += WS_ROWS) {
        const bool more = k + WS_ROWS < k1;
        if (more) request(k + WS_ROWS);                     // in flight under this batch's products
        const float* ap = As + g * WS_P + 16 * w + n;
        const float* bp = Bs + g * WS_P + 4 * n;
#pragma unroll
        for (int st = 0; st < WS_ROWS / 4; ++st) {
            const float a = ap[4 * st * WS_P];
            const float4 b0 = *reinterpret_cast<const float4*>(bp + 4 * st * WS_P);
            const float4 b1 = *reinterpret_cast<const float4*>(bp + 4 * st * WS_P + 64);
            acc[0] = mfma4(a, b0.x, acc[0]); acc[1] = mfma4(a, b0.y, acc[1]);
            acc[2] = mfma4(a, b0.z, acc[2]); acc[3] = mfma4(a, b0.w, acc[3]);
            acc[4] = mfma4(a, b1.x, acc[4]); acc[5] = mfma4(a, b1.y, acc[5]);
            acc[6] = mfma4(a, b1.z, acc[6]); acc[7] = mfma4(a, b1.w, acc[7]);
        }
        if (more) {
            __syncthreads();                                // every wave is done with this batch
            deposit(k + WS_ROWS);
            __syncthreads();
        }
    }
    float* out = p.part + ((long)(b * p.splits + s) * p.M + im * 128 + 16 * w + 4 * g) * p.ldc + in * 128 + 4 * n;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        st4(out + r * p.ldc, make_float4(p.alpha * acc[0][r], p.alpha * acc[1][r], p.alpha * acc[2][r], p.alpha * acc[3][r]));
        st4(out + r * p.ldc + 64, make_float4(p.alpha * acc[4][r], p.alpha * acc[5][r], p.alpha * acc[6][r], p.alpha * acc[7][r]));
    }
}

}  // namespace

// token rows per split: 320 (what the layer's split-K sink is sized for: xlnet_layer.hip splitk_sink_floats), more when that
// would be more than 128 splits; a multiple of the batch
void t4r_wgrad_stream_plan(int K, int* splits, int* kper) {
    int kp = 320;
    if ((K + kp - 1) / kp > 128) kp = ((K + 127) / 128 + WS_ROWS - 1) / WS_ROWS * WS_ROWS;
    *kper = kp;
    *splits = (K + kp - 1) / kp;
}

// Is this accumulating  C += A^T B  one for the streaming kernel?  (whole 128 x 128 blocks, 16-byte loadable operands, a
// contraction long enough to be split)
bool t4r_wgrad_stream_ok(const GemmParams& p) {
    static const int on = [] { const char* e = getenv("T4R_WGRAD_STREAM"); return e ? atoi(e) : 0; }();
    return on && p.M % 128 == 0 && p.N % 128 == 0 && p.K >= 1280 && p.vecA && p.vecB && p.accumulate && p.epilogue == EPI_NONE &&
           !p.sg_lse && !p.rk_thr && p.ldc == p.N;
}

// part: the sink's partial images [batch][splits][M][ldc] (p.splitk splits registered by the caller)
int t4r_wgrad_stream_launch(const GemmParams& p, int batch, int kper, float* part, hipStream_t st) {
    WgradStream q;
    q.A = p.A; q.B = p.B; q.part = part; q.lda = p.lda; q.ldb = p.ldb; q.ldc = p.ldc; q.sA = p.sA; q.sB = p.sB;
    q.M = p.M; q.K = p.K; q.kper = kper; q.splits = p.splitk; q.mb = p.M / 128; q.nb = p.N / 128; q.alpha = p.alpha;
    const size_t smem = (size_t)2 * WS_ROWS * WS_P * sizeof(float);
    static T4rLdsAttr once;
    t4r_ensure_dynamic_lds((const void*)wgrad_stream_kernel, smem, once);
    const long grid = (long)batch * q.splits * q.mb * q.nb;
    hipLaunchKernelGGL(wgrad_stream_kernel, dim3((unsigned)grid), dim3(512), smem, st, q);
    T4R_LAUNCH_CHECK();
    return 0;
}
