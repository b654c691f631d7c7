// Host side of the GEMM: tile / split-K / precision selection and the C ABI entry points.
// The kernel template lives in gemm_kernel.h (shared with gemm_half.hip, which instantiates the
// bf16 / fp16 matrix-core variants in its own translation unit).
#include "gemm_kernel.h"
#include <atomic>

template <int BM, int BN, int BK, bool TA, bool TB, int FEAT>
static int launch_feat(const GemmParams& p, int batch, hipStream_t stream) {
    if (p.vecA && p.vecB) return launch_vec<BM, BN, BK, TA, TB, FEAT, true>(p, batch, stream);
    return launch_vec<BM, BN, BK, TA, TB, FEAT, false>(p, batch, stream);
}

template <int BM, int BN, int BK, bool TA, bool TB>
static int launch_cfg(const GemmParams& p, int batch, hipStream_t stream) {
    if (p.sg_lse) {
        // softmax-gradient operand: only the tile the head uses is instantiated
        if (BM == 64 && BN == 64 && !TB) return launch_feat<64, 64, BK, TA, false, 1>(p, batch, stream);
        t4r_set_error("gemm: softmax-grad operand needs the 64x64 tile and transB = 0");
        return -1;
    }
    if (p.rk_thr) {
        if (BM == 64 && BN == 64 && !TA && TB && p.splitk == 1) return launch_feat<64, 64, BK, false, true, 4>(p, batch, stream);
        t4r_set_error("gemm: the rank epilogue needs the 64x64 NT tile without split-K");
        return -1;
    }
    if (p.drop.p > 0.f && (p.epilogue == EPI_BIAS_GELU || p.epilogue == EPI_BIAS_RESID)) {
        if (BM == 64 && BN == 64) return launch_feat<64, 64, BK, TA, TB, 2>(p, batch, stream);
        t4r_set_error("gemm: epilogue dropout needs the 64x64 tile");
        return -1;
    }
    return launch_feat<BM, BN, BK, TA, TB, 0>(p, batch, stream);
}

// ---- precision of the contraction (process-wide; the backward runs on autograd's thread, so not thread-local)
//   0 fp32 matrix cores (the reference's own arithmetic)
//   1 fp32-accurate on the bf16 matrix cores (exact 3-way split, six products) wherever the operands allow it
//   2 / 3 mixed precision, bf16 / fp16 operands, fp32 accumulation (the reference's AMP: trainer.py:363-367)
//   4 auto (default): fp32 accuracy, the split form on the shapes where it measured faster (see auto_split)
static std::atomic<int> g_prec{-1};
// 1 in an experiment build (-DT4R_EXPERIMENTAL): the host side reads its A/B switches only then (transformers4rec_amd/_lib.py: exp_env)
extern "C" int t4r_experimental_build(void) {
#ifdef T4R_EXPERIMENTAL
    return 1;
#else
    return 0;
#endif
}
extern "C" void t4r_set_precision(int mode) { g_prec.store(mode < 0 || mode > 4 ? 0 : mode); }
extern "C" int t4r_get_precision(void) {
    int m = g_prec.load();
    if (m < 0) {
        const char* e = getenv("T4R_GEMM_PREC");
        m = e ? atoi(e) : 4;        // default: auto (fp32 accuracy; the split form where it is faster)
        if (m < 0 || m > 4) m = 4;
        g_prec.store(m);
    }
    return m;
}
int t4r_gemm_half_dispatch(const GemmParams& p, int batch, int ta, int tb, int big, int prec, hipStream_t stream);
int t4r_tok_gemm_try(const GemmParams& p, int batch, int ta, int tb, hipStream_t stream);     // tok_gemm.hip

// shapes on which the split form beat the fp32 matrix cores (tools/gemm_bench.py --prec, profiles/r02_*)
static bool auto_split(const GemmParams& p, bool ta, bool tb) {
    static long min_flops = -1;
    if (min_flops < 0) { const char* e = t4r_exp_getenv("T4R_GEMM_AUTO_MIN_GFLOP"); min_flops = (e ? atol(e) : 2) * 1000000000L; }
    return 2.0 * p.M * p.N * (double)p.K >= (double)min_flops;
}

// ---------------------------------------------------------------- deterministic split-K (two stages)
// Split-K launches normally add their partial tiles into C with fp32 atomics: run-to-run non-deterministic, and every
// split of a tile finishes at the same time and hits the same addresses (same-address atomics serialise at the memory
// side: a 128 x 128-tile experiment, tools/wgrad_planes_experiment.hip, lost 10-20 of 33-43 us to them).  While a SINK
// is installed (t4r_splitk_sink_begin: the XLNet layer backward does, with a slice of its scratch), accumulating
// split-K launches store their partial tiles into the sink instead and register a job; ONE launch
// (t4r_splitk_sink_flush) then adds the partials of every job in split order:  C += ((p0 + p1) + p2) + ...
// The second stages of the layer's column reductions (bias / LayerNorm / attention-bias gradients) ride in the same
// launch (t4r_splitk_sink_add_reduce): one reduction launch per layer backward instead of four small ones + atomics.
// Measured in the step at BASELINE configs[1] (single stream, rocprofv3): FF weight gradients 33.8 -> 32.3 us, the
// D x D ones 24.2 -> 20.5 us, the reduction 11 us per layer: 560 -> 549 us per step.  The point is the fixed order:
// the body's parameter gradients are now bit-reproducible run to run (tests/test_kernels_gpu.py).
struct SplitKJob { const float* part; float* out; int n4; int splits; long stride4; int accumulate; };
constexpr int kMaxSplitKJobs = 20;
struct SplitKJobs { SplitKJob j[kMaxSplitKJobs]; int n; int blk_end[kMaxSplitKJobs]; };
struct SplitKSink { float* ws = nullptr; long cap = 0, used = 0; SplitKJobs jobs; bool on = false; int bypassed = 0; };
static thread_local SplitKSink g_sink;

// workgroup = 16 float4 columns x 16 split groups; group g adds splits g, g + 16, ... in order (four loads in flight per
// thread: a job with 256 partial rows and 32 columns is ONE workgroup, its time is its dependent load chain), then the
// groups are added in order
__device__ __forceinline__ void add4(float4& a, const float4 v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
__global__ __launch_bounds__(256) void splitk_reduce_kernel(SplitKJobs jobs) {
    __shared__ float4 sm[16][16];
    int ji = 0;
    while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.blk_end[ji]) ++ji;
    const SplitKJob job = jobs.j[ji];
    const int b = blockIdx.x - (ji ? jobs.blk_end[ji - 1] : 0);
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int i = b * 16 + c;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < job.n4) {
        const float4* p = reinterpret_cast<const float4*>(job.part) + i;
        int s = g;
        for (; s + 48 < job.splits; s += 64) {
            const float4 v0 = p[(long)s * job.stride4], v1 = p[(long)(s + 16) * job.stride4];
            const float4 v2 = p[(long)(s + 32) * job.stride4], v3 = p[(long)(s + 48) * job.stride4];
            add4(acc, v0); add4(acc, v1); add4(acc, v2); add4(acc, v3);
        }
        for (; s < job.splits; s += 16) add4(acc, p[(long)s * job.stride4]);
    }
    sm[g][c] = acc;
    __syncthreads();
    if (g == 0 && i < job.n4) {
#pragma unroll
        for (int r = 1; r < 16; ++r) add4(acc, sm[r][c]);
        float4* o = reinterpret_cast<float4*>(job.out) + i;
        if (job.accumulate) add4(acc, *o);
        *o = acc;
    }
}

void t4r_splitk_sink_begin(float* ws, long cap_floats) {
    static const int enabled = [] { const char* e = t4r_exp_getenv("T4R_SPLITK_SINK"); return e ? atoi(e) : 1; }();
    g_sink.ws = ws; g_sink.cap = cap_floats; g_sink.used = 0; g_sink.jobs.n = 0; g_sink.on = enabled && ws && cap_floats > 0;
    g_sink.bypassed = 0;
}
int t4r_splitk_sink_flush(hipStream_t st) {
    SplitKJobs& J = g_sink.jobs;
    if (g_sink.on && J.n > 0) {
        int blocks = 0;
        for (int i = 0; i < J.n; ++i) { blocks += (J.j[i].n4 + 15) / 16; J.blk_end[i] = blocks; }
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, J);
        T4R_LAUNCH_CHECK();
    }
    J.n = 0;       // the partial buffers stay allocated until the sink ends: a later flush may run on another stream
    return 0;
}
void t4r_splitk_sink_end() { g_sink.on = false; g_sink.ws = nullptr; g_sink.cap = 0; g_sink.jobs.n = 0; }
// The same mechanism for a caller outside the layer (C ABI, include/t4r_hip.h): between _begin and _end every ACCUMULATING
// split-K t4r_gemm_f32 of this thread stores its partial tiles into `ws` instead of adding them with atomics; _flush adds
// them in split order (one launch).  The input block's projection weight gradient uses it (features.py): it was the one
// order-dependent sum of a BASELINE configs[2] training step.
extern "C" void t4r_gemm_splitk_sink_begin(float* ws, long cap_floats) { t4r_splitk_sink_begin(ws, cap_floats); }
extern "C" int t4r_gemm_splitk_sink_flush(void* stream) { return t4r_splitk_sink_flush((hipStream_t)stream); }
extern "C" void t4r_gemm_splitk_sink_end(void) { t4r_splitk_sink_end(); }
// split-K launches since the last _begin of this thread that were ELIGIBLE for the sink but did not get room in it (too small
// a workspace, too many jobs) and therefore added their partial tiles with fp32 atomics: correct, but not bit-reproducible.
// A caller that sized the workspace itself asks here instead of trusting its copy of the launcher's split rule.
extern "C" int t4r_gemm_splitk_sink_bypassed(void) { return g_sink.bypassed; }

// a split-K launch asks for room: returns the partial buffer (and registers the jobs) or null (-> atomics)
static float* splitk_sink_take(const GemmParams& p, int batch) {
    SplitKSink& k = g_sink;
    if (!k.on || !p.accumulate || p.epilogue != EPI_NONE || p.sg_lse || p.rk_thr) return nullptr;
    const long n = (long)p.M * p.ldc;                       // one partial = C's [M][ldc] image (dense outputs: ldc == N)
    if (p.ldc != p.N || n % 4 || ((uintptr_t)p.C & 15) || (p.sC % 4)) return nullptr;        // not a shape the sink takes
    if (k.jobs.n + batch > kMaxSplitKJobs) { ++k.bypassed; return nullptr; }
    const long need = n * p.splitk * batch;
    if (k.used + need > k.cap || n / 4 > 0x7fffffffL) { ++k.bypassed; return nullptr; }
    float* part = k.ws + k.used;
    k.used += need;
    for (int b = 0; b < batch; ++b)
        k.jobs.j[k.jobs.n++] = SplitKJob{part + (long)b * p.splitk * n, p.C + b * p.sC, (int)(n / 4), p.splitk, n / 4, 1};
    return part;
}
// the second stage of a column reduction (elementwise.hip: t4r_reduce_partials_launch, out_s[i] (+)= sum_b part[b * n + off_s + i])
// joins the same launch while a sink is installed: false -> the caller launches its own kernel
bool t4r_splitk_sink_add_reduce(const float* part, int nblocks, int n, float* const* outs, const int* lens, const int* accs,
                                int n_seg) {
    SplitKSink& k = g_sink;
    if (!k.on || nblocks <= 0 || (n & 3) || ((uintptr_t)part & 15)) return false;
    int live = 0;
    for (int s = 0; s < n_seg; ++s) {
        if (lens[s] & 3) return false;
        // only sums ACCUMULATED into their output are deferred (parameter gradients, read after the layer); an overwriting
        // one is an intermediate of the layer (d k_r with a shared k_r feeds the r weight gradient) and runs at once
        if (outs[s] && lens[s] > 0) { if (((uintptr_t)outs[s] & 15) || !accs[s]) return false; ++live; }
    }
    if (k.jobs.n + live > kMaxSplitKJobs) return false;
    int off = 0;
    for (int s = 0; s < n_seg; ++s) {
        if (outs[s] && lens[s] > 0) k.jobs.j[k.jobs.n++] = SplitKJob{part + off, outs[s], lens[s] / 4, nblocks, n / 4, accs[s]};
        off += lens[s];
    }
    return true;
}

// operand maxima of the NEXT launch (device words: bits of max |A|, max |B|): a launch that would run in the three-plane
// bf16 form runs in the two-way fp16 form instead (gemm_kernel.h: PREC 4).  Consumed by that launch.
static thread_local const float* g_amax_a = nullptr;
static thread_local const float* g_amax_b = nullptr;
static thread_local int g_amax_n = 0, g_amax_nb = 0;
void t4r_gemm_operand_amax(const float* a, const float* b, int n) { g_amax_a = a; g_amax_b = b; g_amax_n = g_amax_nb = n; }
void t4r_gemm_operand_amax2(const float* a, int na, const float* b, int nb) { g_amax_a = a; g_amax_b = b; g_amax_n = na; g_amax_nb = nb; }

#ifdef T4R_EXPERIMENTAL     /* tools/experimental/wgrad_stream.hip: the K-streaming weight-gradient kernel (measured slower inside the step) */
void t4r_wgrad_stream_plan(int K, int* splits, int* kper);
bool t4r_wgrad_stream_ok(const GemmParams& p);
int t4r_wgrad_stream_launch(const GemmParams& p, int batch, int kper, float* part, hipStream_t st);
// tools/experimental/wgrad_units.hip: read-once, cut-once 128 x 128 units in the two-way fp16 form (T4R_WGRAD_UNITS=1)
void t4r_wgrad_units_plan(int K, int* splits, int* kper);
bool t4r_wgrad_units_ok(const GemmParams& p);
int t4r_wgrad_units_launch(const GemmParams& p, int batch, int kper, float* part, hipStream_t st);
#endif

template <bool TA, bool TB>
static int launch_layout(GemmParams& p, int batch, int splitk_req, hipStream_t stream) {
    const float* amax_a = g_amax_a;
    const float* amax_b = g_amax_b;
    const int amax_n = g_amax_n, amax_nb = g_amax_nb;
    g_amax_a = g_amax_b = nullptr;
    p.amaxA = p.amaxB = nullptr; p.n_amax = p.n_amax_b = 0;
#ifdef T4R_EXPERIMENTAL
    // long-K weight gradients with a split-K sink installed (the XLNet layer backward): the K-streaming kernel (wgrad_stream.hip)
    if (TA && !TB && splitk_req < 0 && g_sink.on && t4r_wgrad_stream_ok(p)) {
        int splits = 1, kper = 0;
        t4r_wgrad_stream_plan(p.K, &splits, &kper);
        p.splitk = splits;
        float* part = splitk_sink_take(p, batch);
        if (part) return t4r_wgrad_stream_launch(p, batch, kper, part, stream);
        p.splitk = 1;
    }
    if (TA && !TB && splitk_req < 0 && g_sink.on && amax_a && amax_b) {
        p.amaxA = amax_a; p.amaxB = amax_b; p.n_amax = amax_n; p.n_amax_b = amax_nb;
        if (t4r_wgrad_units_ok(p)) {
            int splits = 1, kper = 0;
            t4r_wgrad_units_plan(p.K, &splits, &kper);
            p.splitk = splits;
            float* part = splitk_sink_take(p, batch);
            if (part) return t4r_wgrad_units_launch(p, batch, kper, part, stream);
            p.splitk = 1;
        }
        p.amaxA = p.amaxB = nullptr; p.n_amax = p.n_amax_b = 0;
    }
#endif
    // tokens x small weight in an fp32-accurate mode: the token-stationary kernel (operands cut once, tok_gemm.hip)
    if (!TA && (splitk_req == 0 || splitk_req == 1)) {
        const int mode = t4r_get_precision();
        if (mode == 4 || mode == 1) {
            p.splitk = 1;
            const int rc = t4r_tok_gemm_try(p, batch, TA, TB, stream);
            if (rc) return rc < 0 ? rc : 0;
        }
    }
    // tile choice: prefer 128x128; drop to 64-wide tiles when the grid would not fill 256 CUs
    auto nblk = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * batch; };
    // Measured on MI355X (tools/gemm_bench.py, profiles/r01_b_gemm_tile_sweep.txt): on the layer shapes
    // and the head's backward products the 64x64x16 tile (7-8 waves/SIMD resident) beats the 128-wide
    // tiles -- latency- not LDS-bound, more workgroups in flight win (head dW 47 -> 79 TF/s).  The one
    // exception is chosen by shape below.
    int bm = 64, bn = 64;
    static int tile_sel = -1;
    if (tile_sel < 0) { const char* e = t4r_exp_getenv("T4R_GEMM_TILE"); tile_sel = e ? atoi(e) : 0; }
    if (tile_sel == 1) { bm = 64; bn = 128; } else if (tile_sel == 2) { bm = 128; bn = 64; }
    else if (tile_sel == 3) { bm = 64; bn = 64; } else if (tile_sel == 4) { bm = 128; bn = 128; }
    else if (!TA && TB && p.M >= 1024 && p.N >= 32768 && !p.sg_lse && !p.rk_thr && p.epilogue == EPI_NONE) {
        // the vocabulary-wide logits product (end-of-round pipeline): per output the workgroup pulls half
        // as much of X and W through L2 with a 128 x 128 tile, 759 vs 797 us stand-alone at C2
        bm = 128; bn = 128;
    }
    static int bk_sel = -1;
    if (bk_sel < 0) { const char* e = t4r_exp_getenv("T4R_GEMM_BK"); bk_sel = e ? atoi(e) : 0; }
    // precision of this launch: the half-precision variants need 16-byte loadable operands; the rank epilogue
    // (exact ranks of the evaluation head) always stays on the fp32 matrix cores
    int half_big = 0;
    int prec = t4r_get_precision();
    if (prec == 4) prec = auto_split(p, TA, TB) ? 1 : 0;
    if (prec && (!(p.vecA && p.vecB) || p.rk_thr)) prec = 0;
    if (prec && p.sg_lse && TB) prec = 0;
    if (prec) {
        static int half_tile = -1;
        if (half_tile < 0) { const char* e = t4r_exp_getenv("T4R_GEMM_HALF_TILE"); half_tile = e ? atoi(e) : 0; }
        // 128 x 128 only with one operand plane (the three-plane images of a 128 x 128 tile take 101 KB of LDS)
        const bool feat = p.sg_lse || (p.drop.p > 0.f && (p.epilogue == EPI_BIAS_GELU || p.epilogue == EPI_BIAS_RESID));
        int big = (!feat && prec >= 2 && (half_tile == 4 || (half_tile == 0 && bm == 128 && bn == 128))) ? 1 : 0;
        // experiment knobs for the three-plane form: T4R_GEMM_SPLIT_TILE = 4 (128 x 128, one workgroup per CU) | 2 (128 x 64)
        static int split_tile = -1;
        if (split_tile < 0) { const char* e = t4r_exp_getenv("T4R_GEMM_SPLIT_TILE"); split_tile = e ? atoi(e) : 0; }
        if (!feat && prec == 1 && (split_tile == 4 || split_tile == 2) && p.M >= 256 && p.N >= 128) big = split_tile == 4 ? 1 : 2;
        // measured (profiles/r02_b_gemm_prec_bench.txt): the 128 x 64 tile (half the A-operand LDS reads per MFMA)
        // wins the large plain products (C5 body 2459 -> 2233 us, square 4096 1051 -> 944 us, logits 700 -> 693 us)
        // and loses or ties below ~20 GFLOP
        if (!feat && prec == 1 && split_tile == 0 && p.M >= 1024 && p.N >= 512 && 2.0 * p.M * p.N * (double)p.K >= 2e10) big = 2;
        // experiment: the softmax-gradient products of the head on the 128 x 64 tile (T4R_GEMM_SG_TILE=2)
        static int sg_tile = -1;
        if (sg_tile < 0) { const char* e = t4r_exp_getenv("T4R_GEMM_SG_TILE"); sg_tile = e ? atoi(e) : 0; }
        if (p.sg_lse && prec == 1 && sg_tile == 2 && p.M >= 1024) big = 2;
        bm = big ? 128 : 64;
        bn = big == 1 ? 128 : 64;
        half_big = big;
        if (prec == 1 && amax_a && amax_b && !feat && p.epilogue == EPI_NONE) {
            prec = 4; p.amaxA = amax_a; p.amaxB = amax_b; p.n_amax = amax_n; p.n_amax_b = amax_nb;
            bm = bn = 64; half_big = 0;
        }
    }
    int splitk = splitk_req;
    if (splitk_req == 0) {  // auto: only when the caller allows atomics (accumulating outputs)
        splitk = 1;
    } else if (splitk_req < 0) {
        const long blocks = nblk(bm, bn);
        const int bk0 = prec ? 32 : (bk_sel ? bk_sel : 16);
        const int kt = (p.K + bk0 - 1) / bk0;
        // enough workgroups to fill 256 CUs x 8 (the k-loop of one workgroup hides latency only
        // through other resident workgroups), but at least ~20 k-tiles each so that the atomics of
        // the epilogue stay a small part.  Measured (tools/gemm_bench.py): head dX 2765x128x100001
        // 1076 us at 1024 workgroups, 786 us at 4096; wgrad 128x512x20480 best at 64 splits.
        static long target = -1, min_tiles = -1;
        if (target < 0) { const char* e = t4r_exp_getenv("T4R_GEMM_SPLIT_TARGET"); target = e ? atol(e) : 4096; }
        if (min_tiles < 0) { const char* e = t4r_exp_getenv("T4R_GEMM_SPLIT_MIN_TILES"); min_tiles = e ? atol(e) : 20; }
        splitk = (int)max(1L, min((long)kt / min_tiles, target / max(1L, blocks)));
        splitk = min(splitk, 256);
    }
    p.splitk = max(1, splitk);
    p.part = nullptr;
    if (p.splitk > 1) {
        // every split owns at least one k-tile (a partial tile must be written by its split)
        const int bk0 = prec ? 32 : (bk_sel ? bk_sel : 16);
        const int kt = (p.K + bk0 - 1) / bk0, kt_per = (kt + p.splitk - 1) / p.splitk;
        p.splitk = (kt + kt_per - 1) / kt_per;
        if (p.splitk > 1) p.part = splitk_sink_take(p, batch);
    }
    // k-tile depth 16.  BK = 32 (16 MFMAs per barrier) wins isolated long-K launches (square 111 ->
    // 115 TF, wgrads 72 -> 80, head dX 91 -> 96; tools/gemm_bench.py) but loses on the K = 128
    // contractions (logits 85 -> 76 TF) and, selected per launch by K, made the whole training step
    // slower (6.31 vs 6.23 ms, same box): the softmax-gradient variants pay for the doubled staging
    // registers.  T4R_GEMM_BK=32 keeps it available for experiments.
    const int BK = bk_sel ? bk_sel : 16;
    static int xcd_sel = -1;
    if (xcd_sel < 0) { const char* e = t4r_exp_getenv("T4R_GEMM_XCD"); xcd_sel = e ? atoi(e) : 1; }
    p.xcd_order = xcd_sel;
    if (p.splitk > 1) {
        if (p.epilogue != EPI_NONE) { t4r_set_error("gemm: split-K needs epilogue NONE"); return -1; }
        if (!p.accumulate && !p.part) {  // atomics accumulate: start from zero unless the caller accumulates
            for (int b = 0; b < batch; ++b)
                (void)hipMemset2DAsync(p.C + b * p.sC, p.ldc * sizeof(float), 0, p.N * sizeof(float), p.M, stream);
        }
    }
    if (prec) return t4r_gemm_half_dispatch(p, batch, TA, TB, half_big, prec, stream);
    if (BK == 16) {
        if (bm == 128 && bn == 128) return launch_cfg<128, 128, 16, TA, TB>(p, batch, stream);
        if (bm == 64 && bn == 128) return launch_cfg<64, 128, 16, TA, TB>(p, batch, stream);
        if (bm == 128 && bn == 64) return launch_cfg<128, 64, 16, TA, TB>(p, batch, stream);
        return launch_cfg<64, 64, 16, TA, TB>(p, batch, stream);
    }
    if (bm == 128 && bn == 128) return launch_cfg<128, 128, 32, TA, TB>(p, batch, stream);
    if (bm == 64 && bn == 128) return launch_cfg<64, 128, 32, TA, TB>(p, batch, stream);
    if (bm == 128 && bn == 64) return launch_cfg<128, 64, 32, TA, TB>(p, batch, stream);
    return launch_cfg<64, 64, 32, TA, TB>(p, batch, stream);
}

struct SoftmaxGradA { const float* lse; const long* labels; const float* gout; int rows, V; float smooth; int yoff; };
static thread_local const SoftmaxGradA* g_sg = nullptr;   // set only by t4r_gemm_softmax_grad_f32
struct RankEpi { const float* thr; const long* label; int* count; };
static thread_local const RankEpi* g_rank = nullptr;      // set only by t4r_rank_of_target_f32

// Internal C++ entry used by the composite (layer / head) launchers.
int t4r_gemm_launch(hipStream_t stream, int transA, int transB, int M, int N, int K, float alpha,
                    const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                    const float* bias, int epilogue, float* aux, long ldaux, int splitk,
                    int accumulate, int batch, long sA, long sB, long sC, const DropCfg* drop) {
    if (M <= 0 || N <= 0) { g_amax_a = g_amax_b = nullptr; return 0; }     // operand maxima announced for THIS launch die with it
    T4R_CHECK_ARG(K > 0 && A && B && C && batch >= 1, "gemm: bad arguments");
    GemmParams p;
    p.M = M; p.N = N; p.K = K;
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.C = C; p.ldc = ldc;
    p.bias = bias; p.aux = aux; p.ldaux = ldaux; p.alpha = alpha; p.epilogue = epilogue;
    p.accumulate = accumulate; p.sA = sA; p.sB = sB; p.sC = sC;
    p.vecA = ((uintptr_t)A % 16 == 0) && (lda % 4 == 0) && (sA % 4 == 0);
    p.vecB = ((uintptr_t)B % 16 == 0) && (ldb % 4 == 0) && (sB % 4 == 0);
    p.splitk = 1;
    p.part = nullptr;
    p.drop = drop ? *drop : make_drop(0.f, 0, 0);
    p.sg_lse = nullptr; p.sg_labels = nullptr; p.sg_gout = nullptr; p.sg_rows = 1; p.sg_V = 1; p.sg_smooth = 0.f; p.sg_yoff = 0;
    p.rk_thr = nullptr; p.rk_label = nullptr; p.rk_count = nullptr;
    if (g_rank) { p.rk_thr = g_rank->thr; p.rk_label = g_rank->label; p.rk_count = g_rank->count; }
    if (g_sg) {
        p.sg_lse = g_sg->lse; p.sg_labels = g_sg->labels; p.sg_gout = g_sg->gout;
        p.sg_rows = g_sg->rows; p.sg_V = g_sg->V; p.sg_smooth = g_sg->smooth; p.sg_yoff = g_sg->yoff;
    }
    if (transA) {
        if (transB) return launch_layout<true, true>(p, batch, splitk, stream);
        return launch_layout<true, false>(p, batch, splitk, stream);
    }
    if (transB) return launch_layout<false, true>(p, batch, splitk, stream);
    return launch_layout<false, false>(p, batch, splitk, stream);
}

extern "C" int t4r_gemm_f32(void* stream, int transA, int transB, int M, int N, int K, float alpha,
                            const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                            const float* bias, int epilogue, float* aux, long ldaux, int splitk,
                            int accumulate, int batch, long strideA, long strideB, long strideC,
                            float drop_p, unsigned long long seed, unsigned long long ctr_hi) {
    T4R_CHECK_ARG(epilogue != EPI_BIAS_RESID || aux, "gemm: EPI_BIAS_RESID needs the residual in aux");
    const DropCfg dc = make_drop(drop_p, seed, ctr_hi);
    return t4r_gemm_launch((hipStream_t)stream, transA, transB, M, N, K, alpha, A, lda, B, ldb, C,
                           ldc, bias, epilogue, aux, ldaux, splitk, accumulate, batch, strideA,
                           strideB, strideC, drop_p > 0.f ? &dc : nullptr);
}

// Head backward contractions with CrossEntropyLoss' backward fused into the A operand:
//   transA = 0:  C[N_rows, N] (+)= alpha * dlogits[N_rows, V] @ B[V, N]          (d X = dlogits @ W)
//   transA = 1:  C[V, N]     (+)= alpha * dlogits[N_rows, V]^T @ B[N_rows, N]    (d W = dlogits^T @ X)
// where dlogits = (*grad_out / N_rows) * (softmax(logits) - target) is formed from `logits`,
// `lse` and `labels` while the tile is staged.  Replaces model/prediction_task.py:446 (loss
// backward through CrossEntropyLoss) + the autograd of :664.
// logits holds the columns [yoff, yoff + Vc) of the full [n_rows, V] logits (Vc = V, yoff = 0: all of them);
// the label-smoothing term eps / V and the mean 1 / n_rows refer to the full problem.
int t4r_gemm_softmax_grad_launch(hipStream_t stream, int transA, int n_rows, int Vc, int V, int yoff, int N,
                                 float alpha, const float* logits, long ld_logits, const float* lse,
                                 const long* labels, const float* grad_out, float label_smoothing, const float* B,
                                 long ldb, float* C, long ldc, int splitk, int accumulate) {
    T4R_CHECK_ARG(lse && labels && logits, "gemm_softmax_grad: null operand");
    SoftmaxGradA sg{lse, labels, grad_out, n_rows, V, label_smoothing, yoff};
    g_sg = &sg;
    const int M = transA ? Vc : n_rows, K = transA ? n_rows : Vc;
    const int rc = t4r_gemm_launch(stream, transA, 0, M, N, K, alpha, logits, ld_logits, B, ldb,
                                   C, ldc, nullptr, EPI_NONE, nullptr, 0, splitk, accumulate, 1, 0, 0, 0,
                                   nullptr);
    g_sg = nullptr;
    return rc;
}

extern "C" int t4r_gemm_softmax_grad_f32(void* stream, int transA, int n_rows, int V, int N, float alpha,
                                         const float* logits, long ld_logits, const float* lse,
                                         const long* labels, const float* grad_out, float label_smoothing,
                                         const float* B, long ldb, float* C, long ldc, int splitk,
                                         int accumulate) {
    return t4r_gemm_softmax_grad_launch((hipStream_t)stream, transA, n_rows, V, V, 0, N, alpha, logits, ld_logits,
                                        lse, labels, grad_out, label_smoothing, B, ldb, C, ldc, splitk, accumulate);
}

// Fused eval head: rank of the target item among alpha * X @ W^T without materialising
// the [n_rows, V] scores.  target_score[row] must be the row's own score of its label column as THIS
// kernel computes it (run t4r_gemm_f32 on the gathered label rows of W and take the diagonal: an
// output element's bits do not depend on its tile position).  rank[row] (int32) is overwritten with
// #{v : score_v > target or (score_v == target and v < label)}: Recall@k = rank < k,
// NDCG@k = rank < k ? 1 / log2(rank + 2) : 0  (ranking_metric.py:107-147, 242-280 with one relevant item).
extern "C" int t4r_rank_of_target_f32(void* stream, int n_rows, int V, int D, float alpha, const float* X,
                                      long ldx, const float* W, long ldw, const float* target_score,
                                      const long* labels, int* rank) {
    if (n_rows == 0) return 0;
    T4R_CHECK_ARG(target_score && labels && rank, "rank_of_target: null pointer");
    if (hipMemsetAsync(rank, 0, sizeof(int) * (size_t)n_rows, (hipStream_t)stream) != hipSuccess) {
        t4r_set_error("rank_of_target: memset failed");
        return -1;
    }
    RankEpi re{target_score, labels, rank};
    g_rank = &re;
    // C is never written by the rank epilogue; a non-null dummy keeps the argument check happy
    const int rc = t4r_gemm_launch((hipStream_t)stream, 0, 1, n_rows, V, D, alpha, X, ldx, W, ldw,
                                   reinterpret_cast<float*>(rank), V, nullptr, EPI_NONE, nullptr, 0, 1, 0, 1, 0, 0,
                                   0, nullptr);
    g_rank = nullptr;
    return rc;
}
