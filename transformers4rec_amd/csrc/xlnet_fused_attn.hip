// Token-tile-stationary kernels of the ATTENTION half of the XLNet layer (round 3), built from the pieces of
// xlnet_fused.h (three-plane bf16 operands cut once, six-product MFMA chain, transposed products: weight = A operand):
//
//   xlnet_proj_kernel      rows [T, D] -> NM matrices [T, D] each:  q, k, v = h @ W_{q,k,v}   (NM = 3, one launch, the token
//                          tile is read and cut once for the three products), and  k_r = pos_emb(_b) @ r  (NM = 1)
//                          (HF modeling_xlnet.py :251-258 einsum('ibh,hnd->ibnd') x 3, :266 k_head_r)
//   xlnet_oproj_ln_kernel  attn_vec -> o-projection + dropout + residual + LayerNorm -> h1      (HF post_attention :142-152)
//   xlnet_ln1_bwd_kernel   d h1 -> LayerNorm backward -> d attn_out (rows of the o weight gradient) -> d attn_vec = d ao @ o
//                          (+ partial sums of d gamma, d beta)
//   xlnet_dh_kernel        d h += d q @ W_q^T + d k @ W_k^T + d v @ W_v^T    (three products into one accumulator; the
//                          d q / d k / d v tiles take turns in two LDS plane buffers)
//   layer_planes_kernel    one launch cuts all nine weight matrices of a layer into their bf16 planes (xlnet_fused.h)
//
// The attention core itself (scores, relative shift, softmax, dropout, P @ v) stays xlnet_attn_mfma.hip.
// Each replaces one or two launches of the general GEMM + an element-wise / LayerNorm launch of the chain in
// xlnet_layer.hip; what they save is the operand cutting per 64 x 64 tile of the general kernel, the LayerNorm round
// trips and the launch boundaries (measured per layer at the benchmark size, same box: see docs/DESIGN_rounds_1_to_4.md 4.1c).
#include "xlnet_fused.h"

// ---------------------------------------------------------------------------------------------- weight planes of a layer
struct PlaneJob {
    const float* src;      // [rows][cols] fp32
    uint16_t* dst;         // plane 0 of the bf16 destination matrix [3][drows][dcols]
    uint16_t* dst_h;       // plane 0 of the fp16 destination [2][drows][dcols] (may be null)
    const float* scale;    // the source's power-of-two scale (device; fp16 planes)
    int pair0;             // first pair index of this job in the launch
    int dplane;            // destination plane stride (elements), both forms
    short rows, cols;      // source shape
    short dcols;           // destination row pitch
    short r0, c0;          // destination offset of this source
    short transpose;       // dst[r0 + c][c0 + r] = src[r][c]  instead of  dst[r0 + r][c0 + c]
    float* dst32;          // fp32 copy in the destination's orientation (exact-fp32 attention block kernels; may be null)
};
#define T4R_MAX_PLANE_JOBS 52        /* four layers x 13 matrices: 2.9 KB of kernel arguments */
struct PlaneJobs { PlaneJob j[T4R_MAX_PLANE_JOBS]; int n; int total; };

__global__ __launch_bounds__(256) void layer_planes_kernel(PlaneJobs jobs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= jobs.total) return;
    int k = 0;
#pragma unroll 1
    for (int q = 1; q < jobs.n; ++q) k = i >= jobs.j[q].pair0 ? q : k;
    const PlaneJob& jb = jobs.j[k];
    const int li = i - jb.pair0;
    const int r = li / (jb.cols / 2), c = (li % (jb.cols / 2)) * 2;
    const float2 v = *reinterpret_cast<const float2*>(jb.src + (long)r * jb.cols + c);
    uint32_t w[3] = {0u, 0u, 0u}, wh[2] = {0u, 0u};
    if (jb.dst) cut3(v.x, v.y, w);
    if (jb.dst_h) {
        const float sc = *jb.scale;
        cut2h(v.x * sc, v.y * sc, wh);
    }
    const long o_n = (long)(jb.r0 + r) * jb.dcols + jb.c0 + c;
    const long o_t0 = (long)(jb.r0 + c) * jb.dcols + jb.c0 + r, o_t1 = o_t0 + jb.dcols;
    if (jb.dst32) {
        if (!jb.transpose) {
            *reinterpret_cast<float2*>(jb.dst32 + o_n) = v;
        } else {
            jb.dst32[o_t0] = v.x;
            jb.dst32[o_t1] = v.y;
        }
    }
    if (jb.dst) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            uint16_t* d = jb.dst + (long)pl * jb.dplane;
            if (!jb.transpose) {
                *reinterpret_cast<uint32_t*>(d + o_n) = w[pl];
            } else {
                d[o_t0] = (uint16_t)(w[pl] & 0xffffu);
                d[o_t1] = (uint16_t)(w[pl] >> 16);
            }
        }
    }
    if (jb.dst_h) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            uint16_t* d = jb.dst_h + (long)pl * jb.dplane;
            if (!jb.transpose) {
                *reinterpret_cast<uint32_t*>(d + o_n) = wh[pl];
            } else {
                d[o_t0] = (uint16_t)(wh[pl] & 0xffffu);
                d[o_t1] = (uint16_t)(wh[pl] >> 16);
            }
        }
    }
}

// The same planes (fp16 form) from 32 x 32 tiles, one per wave: coalesced 16-byte reads, the cut once, and BOTH orientations written
// as full 64-byte row segments -- the transposed ones through an LDS tile.  layer_planes_kernel above gives every thread two
// source elements and writes a transposed destination as single 2-byte stores 2 * dcols apart: 3.2 M partial-line writes per
// step, 26 us for 12 MB of planes at BASELINE configs[1].  Same bytes out (tests/test_round5_gpu.py compares the two).
// Requirements (checked by the host): fp16 form only (dst == null), rows and cols multiples of 32.
__global__ __launch_bounds__(256) void layer_planes_tiled_kernel(PlaneJobs jobs) {
    constexpr int HP = 40, FP = 36;                      // LDS pitches: 16-bit tile rows (80 B), fp32 tile rows (144 B)
    __shared__ __attribute__((aligned(16))) uint16_t th[4][2][32 * HP];
    __shared__ __attribute__((aligned(16))) float tf[4][32 * FP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_tiles = jobs.total / 512;
    const int gw0 = blockIdx.x * 4 + wave;
    const bool active = gw0 < n_tiles;
    const int gw = active ? gw0 : n_tiles - 1;
    int k = 0;
#pragma unroll 1
    for (int q = 1; q < jobs.n; ++q) k = gw * 512 >= jobs.j[q].pair0 ? q : k;
    const PlaneJob& jb = jobs.j[k];
    const int lt = gw - jb.pair0 / 512, tpr = jb.cols / 32;
    const int tr = (lt / tpr) * 32, tc = (lt % tpr) * 32;          // tile origin in the source
    const int lr = lane >> 3, lc = (lane & 7) * 4;
    const float sc = *jb.scale;
    uint16_t* dh = jb.dst_h;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = lr + 8 * i;
        const float4 v = *reinterpret_cast<const float4*>(jb.src + (long)(tr + r) * jb.cols + tc + lc);
        uint32_t wa[2], wb[2];
        cut2h(v.x * sc, v.y * sc, wa);
        cut2h(v.z * sc, v.w * sc, wb);
        if (!jb.transpose) {
            const long o = (long)(jb.r0 + tr + r) * jb.dcols + jb.c0 + tc + lc;
            if (active) {
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    *reinterpret_cast<uint2*>(dh + (long)pl * jb.dplane + o) = make_uint2(wa[pl], wb[pl]);
                if (jb.dst32) *reinterpret_cast<float4*>(jb.dst32 + o) = v;
            }
        } else {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                uint16_t* t = th[wave][pl];
                t[(lc + 0) * HP + r] = (uint16_t)(wa[pl] & 0xffffu);
                t[(lc + 1) * HP + r] = (uint16_t)(wa[pl] >> 16);
                t[(lc + 2) * HP + r] = (uint16_t)(wb[pl] & 0xffffu);
                t[(lc + 3) * HP + r] = (uint16_t)(wb[pl] >> 16);
            }
            if (jb.dst32) {
                float* t = tf[wave];
                t[(lc + 0) * FP + r] = v.x; t[(lc + 1) * FP + r] = v.y; t[(lc + 2) * FP + r] = v.z; t[(lc + 3) * FP + r] = v.w;
            }
        }
    }
    __syncthreads();
    if (jb.transpose && active) {
        // destination row = source column: lane -> (row cl, half of its 32 entries)
        const int cl = lane >> 1, half = lane & 1;
        const long o = (long)(jb.r0 + tc + cl) * jb.dcols + jb.c0 + tr + 16 * half;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const uint4* t = reinterpret_cast<const uint4*>(th[wave][pl] + cl * HP + 16 * half);
            uint4* d = reinterpret_cast<uint4*>(dh + (long)pl * jb.dplane + o);
            d[0] = t[0];
            d[1] = t[1];
        }
        if (jb.dst32) {
            const float4* t = reinterpret_cast<const float4*>(tf[wave] + cl * FP + 16 * half);
            float4* d = reinterpret_cast<float4*>(jb.dst32 + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = t[e];
        }
    }
}

// max |src| of every source matrix -> its power-of-two scale (one workgroup per source: 16-64 k elements, one coalesced
// pass); a source with `raw` set stores the maximum itself (the bias vector b1)
struct AmaxJob { const float* src; float* out; int n; int raw; };
#define T4R_MAX_AMAX_JOBS 32
struct AmaxJobs { AmaxJob j[T4R_MAX_AMAX_JOBS]; int n; };
__global__ __launch_bounds__(1024) void weight_scales_kernel(AmaxJobs jobs) {
    __shared__ float red[16];
    const AmaxJob jb = jobs.j[blockIdx.x];
    float m = 0.f;
    for (int i = threadIdx.x * 4; i < jb.n; i += 4096) {
        const float4 v = *reinterpret_cast<const float4*>(jb.src + i);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float mm = red[0];
        for (int w = 1; w < 16; ++w) mm = fmaxf(mm, red[w]);
        *jb.out = jb.raw ? mm : pow2_scale(mm);
    }
}

extern "C" long t4r_xlnet_layer_planes_floats(int D) { return layer_planes_floats(D); }

static void add_job(PlaneJobs& js, const float* src, int rows, int cols, const uint16_t* dst, const uint16_t* dst_h,
                    const float* scale, int drows, int dcols, int r0, int c0, int transpose, const float* dst32 = nullptr) {
    // only the form the kernels will read is produced: the fp16 planes (default) or the bf16 planes (T4R_XLNET_FP16X2=0)
    const bool hs = t4r_xlnet_body_fp16x2();
    PlaneJob& j = js.j[js.n++];
    j.src = src; j.dst = hs ? nullptr : const_cast<uint16_t*>(dst); j.dst_h = hs ? const_cast<uint16_t*>(dst_h) : nullptr; j.scale = scale;
    j.rows = (short)rows; j.cols = (short)cols; j.dcols = (short)dcols; j.dplane = drows * dcols;
    j.r0 = (short)r0; j.c0 = (short)c0; j.transpose = (short)transpose; j.pair0 = js.total;
    j.dst32 = const_cast<float*>(dst32);
    js.total += rows * cols / 2;
}
static void add_amax(AmaxJobs& aj, const float* src, int n, const float* out, int raw) {
    aj.j[aj.n++] = AmaxJob{src, const_cast<float*>(out), n, raw};
}

// params: host array of the layer's 15 device pointers in the order of t4r_xlnet_layer_fwd (q, k, v, o, r, ..., W1 at 9,
// b1 at 10, W2 at 11); any of the attention weights may be NULL (feed-forward planes only: t4r_xlnet_ff_prepare).
// Every matrix gets its three bf16 planes and its two fp16 planes (scale of the source from the scales launch before).
static void add_layer_jobs(PlaneJobs& js, AmaxJobs& aj, const float* q, const float* k, const float* v, const float* o,
                           const float* r, const float* W1, const float* b1, const float* W2, int D, float* planes) {
    const LayerPlanes P = carve_planes(planes, D);
    const LayerPlanesH H = carve_planes_h(planes, D);
    const LayerPlanes32 F = carve_planes_32(planes, D);
    const float* sc = H.scale;
    const float* z[3] = {q, k, v};
    for (int i = 0; i < 3; ++i) {
        if (!z[i]) continue;
        add_amax(aj, z[i], D * D, sc + HS_Q + i, 0);
        add_job(js, z[i], D, D, P.QKVT, H.QKVT, sc + HS_Q + i, 3 * D, D, i * D, 0, 1, F.QKVT);
        add_job(js, z[i], D, D, P.QKVN, H.QKVN, sc + HS_Q + i, D, 3 * D, 0, i * D, 0);
    }
    if (r) {
        add_amax(aj, r, D * D, sc + HS_R, 0);
        add_job(js, r, D, D, P.RT, H.RT, sc + HS_R, D, D, 0, 0, 1);
    }
    if (o) {
        add_amax(aj, o, D * D, sc + HS_O, 0);
        add_job(js, o, D, D, P.ON, H.ON, sc + HS_O, D, D, 0, 0, 0);
        add_job(js, o, D, D, P.OT, H.OT, sc + HS_O, D, D, 0, 0, 1, F.OT);
    }
    if (W1) {
        add_amax(aj, W1, 4 * D * D, sc + HS_W1, 0);
        add_job(js, W1, 4 * D, D, P.W1p, H.W1p, sc + HS_W1, 4 * D, D, 0, 0, 0);
        add_job(js, W1, 4 * D, D, P.W1Tp, H.W1Tp, sc + HS_W1, D, 4 * D, 0, 0, 1);
    }
    if (W2) {
        add_amax(aj, W2, 4 * D * D, sc + HS_W2, 0);
        add_job(js, W2, D, 4 * D, P.W2p, H.W2p, sc + HS_W2, D, 4 * D, 0, 0, 0);
        add_job(js, W2, D, 4 * D, P.W2Tp, H.W2Tp, sc + HS_W2, 4 * D, D, 0, 0, 1);
    }
    if (b1) add_amax(aj, b1, 4 * D, sc + HS_B1, 1);
}
static int launch_jobs(hipStream_t st, const PlaneJobs& js, const AmaxJobs& aj) {
    if (js.total == 0) return 0;
    if (aj.n > 0) hipLaunchKernelGGL(weight_scales_kernel, dim3(aj.n), dim3(1024), 0, st, aj);
    // the tiled form takes the fp16 planes of matrices whose sides are multiples of 32 with 16-byte-aligned destinations
    const char* tiled_env = t4r_exp_getenv("T4R_PLANES_TILED");        // read per call (two launches per step): a test flips it in-process
    bool tiled = !tiled_env || atoi(tiled_env) != 0;
    for (int i = 0; i < js.n && tiled; ++i) {
        const PlaneJob& j = js.j[i];
        tiled = !j.dst && j.dst_h && j.rows % 32 == 0 && j.cols % 32 == 0 && j.dcols % 8 == 0 && j.r0 % 32 == 0 && j.c0 % 32 == 0 &&
                ((uintptr_t)j.src & 15) == 0 && ((uintptr_t)j.dst_h & 15) == 0 && (j.dplane % 8) == 0 && ((uintptr_t)j.dst32 & 15) == 0;
    }
    if (tiled) {
        hipLaunchKernelGGL(layer_planes_tiled_kernel, dim3((unsigned)((js.total / 512 + 3) / 4)), dim3(256), 0, st, js);
        T4R_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(layer_planes_kernel, dim3((unsigned)((js.total + 255) / 256)), dim3(256), 0, st, js);
    T4R_LAUNCH_CHECK();
    return 0;
}
// params: host array of the layer's 15 device pointers in the order of t4r_xlnet_layer_fwd (q, k, v, o, r, ..., W1 at 9,
// W2 at 11); any of the attention weights may be NULL (feed-forward planes only: t4r_xlnet_ff_prepare)
static int prepare_launch(hipStream_t st, const float* q, const float* k, const float* v, const float* o, const float* r,
                          const float* W1, const float* b1, const float* W2, int D, float* planes) {
    PlaneJobs js;
    AmaxJobs aj;
    js.n = 0; js.total = 0; aj.n = 0;
    add_layer_jobs(js, aj, q, k, v, o, r, W1, b1, W2, D, planes);
    return launch_jobs(st, js, aj);
}

extern "C" int t4r_xlnet_fused_supported(int D);

// cuts every weight matrix of one layer into its bf16 planes: ONE launch per layer call (planes:
// t4r_xlnet_layer_planes_floats(D) floats).  params: host array of 15 device pointers, order of t4r_xlnet_layer_fwd.
extern "C" int t4r_xlnet_layer_prepare(void* stream, const float* const* params, int D, float* planes) {
    T4R_CHECK_ARG(t4r_xlnet_fused_supported(D), "xlnet_layer_prepare: d_model must be 32, 64 or 128");
    T4R_CHECK_ARG(params && planes, "xlnet_layer_prepare: null pointer");
    return prepare_launch((hipStream_t)stream, params[0], params[1], params[2], params[3], params[4], params[9], params[10],
                          params[11], D, planes);
}
// the feed-forward planes only (stand-alone use of t4r_xlnet_ff_fwd / _bwd); same buffer layout and size
extern "C" int t4r_xlnet_ff_prepare(void* stream, const float* W1, const float* b1, const float* W2, int D, float* planes) {
    T4R_CHECK_ARG(t4r_xlnet_fused_supported(D), "xlnet_ff_prepare: d_model must be 32, 64 or 128");
    T4R_CHECK_ARG(W1 && b1 && W2 && planes, "xlnet_ff_prepare: null pointer");
    return prepare_launch((hipStream_t)stream, nullptr, nullptr, nullptr, nullptr, nullptr, W1, b1, W2, D, planes);
}
extern "C" long t4r_xlnet_ff_planes_floats(int D) { return layer_planes_floats(D); }

// ---------------------------------------------------------------------------------------------- shared device pieces
// rows t0 .. t0 + RT - 1 of src [T, D] (clamped to T - 1) -> three bf16 planes [3][RT][PH] in LDS
template <int D, int RT, int NT, int PH>
__device__ __forceinline__ void tile_to_planes(const float* __restrict__ src, long t0, long T, uint16_t* sh, int tid) {
    constexpr int PLN = RT * PH;
    for (int i = tid; i < RT * (D / 4); i += NT) {
        const int row = i / (D / 4), c4 = (i % (D / 4)) * 4;
        const long t = min(t0 + row, T - 1);
        const float4 v = ld4(src + t * D + c4);
        uint32_t w0[3], w1[3];
        cut3(v.x, v.y, w0);
        cut3(v.z, v.w, w1);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<uint2*>(sh + pl * PLN + row * PH + c4) = make_uint2(w0[pl], w1[pl]);
    }
}

// the same tile as two-way fp16 planes [2][RT][PH], every token row positioned by its OWN power-of-two scale (max |row|
// -> [2^13, 2^14)): rows of very different magnitude in one tile (gradient rows) keep their full 22 bits each.
// sh_inv[row] = 1 / scale (exact), multiplied back in the product's epilogue.  The D / 4 threads of a row are consecutive
// lanes of one wave (D / 4 <= 32).
// `gen` (round 6, the positional keys of a training forward): the rows are not read from src but MADE here -- row t =
// base[t % period] (the [2L, D] positional encoding, cache resident) times the pos_emb dropout mask of element (t, c) (HF
// modeling_xlnet.py:1143: one mask per session, drawn once per forward) -- and written to gen->out [T, D] for the backward's d r
// contraction: the separate element-wise launch that materialised them (and this kernel's 21 MB read of its result) are gone.
struct GenRows { const float* base; long period; DropCfg drop; float* out; };
template <int D, int RT, int NT, int PH>
__device__ __forceinline__ void tile_to_planes_h(const float* __restrict__ src, long t0, long T, uint16_t* sh, float* sh_inv,
                                                 int tid, const GenRows* gen = nullptr) {
    constexpr int PLN = RT * PH, G = D / 4;
    for (int i = tid; i < ((RT * G + NT - 1) / NT) * NT; i += NT) {      // whole waves stay in the loop: the shuffles need their partners
        const bool live = i < RT * G;
        const int row = live ? i / G : RT - 1, c4 = live ? (i % G) * 4 : 0;
        const long t = min(t0 + row, T - 1);
        float4 v;
        if (gen) {                      // workgroup-uniform
            v = ld4(gen->base + (t % gen->period) * D + c4);
            const float4 mk = drop_scale4(gen->drop, (unsigned long long)t * D + c4);
            v.x *= mk.x; v.y *= mk.y; v.z *= mk.z; v.w *= mk.w;
            if (live && t0 + row < T) st4(gen->out + t * D + c4, v);
        } else {
            v = ld4(src + t * D + c4);
        }
        float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        const float sc = pow2_scale(m);
        if (live) {
            uint32_t w0[2], w1[2];
            cut2h(v.x * sc, v.y * sc, w0);
            cut2h(v.z * sc, v.w * sc, w1);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                *reinterpret_cast<uint2*>(sh + pl * PLN + row * PH + c4) = make_uint2(w0[pl], w1[pl]);
            if (c4 == 0) sh_inv[row] = 1.f / sc;
        }
    }
}

// ---------------------------------------------------------------------------------------------- projections
struct ProjParams {
    const float* in;          // [T, D]
    const uint16_t* w[4];     // per output matrix: plane 0 of its weight planes [D][D] (rows = output feature)
    long wpl;                 // plane stride of those planes (elements)
    float* out[4];            // per output matrix: [T, D]
    long T;
    const float* wscale[4];   // HS: the matrices' power-of-two scales (device)
    GenRows gen;              // gen.base != NULL: the input rows are generated (see tile_to_planes_h), `in` is unused
};

// HS: the two-way fp16 form (three matrix instructions per k-step instead of six; per-token and per-matrix power-of-two
// scales, see tile_to_planes_h / weight_scales_kernel)
template <int D, int R, int NM, bool HS = false>
__global__ __launch_bounds__(D * 4) void xlnet_proj_kernel(ProjParams p) {
    constexpr int NW = D / 16, NT = NW * 64, RT = 16 * R, PH = D + 16, PLN = RT * PH;
    extern __shared__ uint16_t smem16[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, g = lane >> 4;
    const long t0 = (long)blockIdx.x * RT;
    float* sh_inv = reinterpret_cast<float*>(smem16 + (HS ? 2 : 3) * PLN);       // HS: [RT] inverse token scales
    if constexpr (HS) tile_to_planes_h<D, RT, NT, PH>(p.in, t0, p.T, smem16, sh_inv, tid, p.gen.base ? &p.gen : nullptr);
    else tile_to_planes<D, RT, NT, PH>(p.in, t0, p.T, smem16, tid);
    __syncthreads();
    const int boff = n * PH + 8 * g;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        f32x4 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = zero4();
        if constexpr (HS) {
            AFragH<D> a;
            load_a2h<D>(a, p.w[m] + (long)(16 * w + n) * D + 8 * g, p.wpl);
            product3h<D, R, PH>(a, smem16 + boff, PLN, acc);
        } else {
            AFrag<D> a;
            load_a3<D>(a, p.w[m] + (long)(16 * w + n) * D + 8 * g, p.wpl);
            product3<D, R, PH>(a, smem16 + boff, PLN, acc);
        }
        const float iw = HS ? 1.f / *p.wscale[m] : 1.f;
        float* o = p.out[m] + 16 * w + 4 * g;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long t = t0 + r * 16 + n;
            const float sc = HS ? sh_inv[r * 16 + n] * iw : 1.f;          // two exact powers of two
            if (t < p.T) st4(o + t * D, make_float4(acc[r][0] * sc, acc[r][1] * sc, acc[r][2] * sc, acc[r][3] * sc));
        }
    }
}

// ---------------------------------------------------------------------------------------------- o-projection + LayerNorm
struct OProjParams {
    const float *av, *h, *gamma, *beta;    // attn_vec [T, D], layer input h [T, D] (residual), LayerNorm parameters
    const uint16_t* planes;                // ON planes [D][D] (three bf16 planes, or the two fp16 planes with *wscale)
    const float* wscale;
    float *ao, *mean, *rstd, *h1;          // saved o-projection output (pre dropout), statistics (all NULL: inference), out
    long T;
    float eps;
    DropCfg drop;
};

template <int D, int R, bool TRAIN, bool HS>
__device__ __forceinline__ void oproj_body(const OProjParams& p, uint16_t* smem16) {
    constexpr int NW = D / 16, NT = NW * 64, RT = 16 * R, PH = D + 16, PLN = RT * PH;
    float* sh_red = reinterpret_cast<float*>(smem16 + 3 * PLN);      // [2][NW][RT]
    float* sh_inv = reinterpret_cast<float*>(smem16 + 2 * PLN);      // HS: [RT] inverse token scales (in the unused third plane)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, g = lane >> 4;
    const long t0 = (long)blockIdx.x * RT;
    f32x4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = zero4();
    if constexpr (HS) {
        tile_to_planes_h<D, RT, NT, PH>(p.av, t0, p.T, smem16, sh_inv, tid);
        AFragH<D> a;
        load_a2h<D>(a, p.planes + (long)(16 * w + n) * D + 8 * g, (long)D * D);
        __syncthreads();
        product3h<D, R, PH>(a, smem16 + n * PH + 8 * g, PLN, acc);
        const float iw = 1.f / *p.wscale;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float sc = sh_inv[r * 16 + n] * iw;
            acc[r][0] *= sc; acc[r][1] *= sc; acc[r][2] *= sc; acc[r][3] *= sc;
        }
    } else {
        tile_to_planes<D, RT, NT, PH>(p.av, t0, p.T, smem16, tid);
        AFrag<D> a;
        load_a3<D>(a, p.planes + (long)(16 * w + n) * D + 8 * g, (long)D * D);
        __syncthreads();
        product3<D, R, PH>(a, smem16 + n * PH + 8 * g, PLN, acc);
    }
    const int f0 = 16 * w + 4 * g;
    const float4 gam = ld4(p.gamma + f0), bet = ld4(p.beta + f0);
    float4 x[R];
    float sum[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long t = t0 + r * 16 + n;
        const long tc = min(t, p.T - 1);
        float4 v = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
        if (TRAIN && t < p.T) st4(p.ao + t * D + f0, v);
        if (TRAIN) {
            const float4 m = drop_scale4(p.drop, (unsigned long long)t * D + f0);
            v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
        }
        const float4 hres = ld4(p.h + tc * D + f0);
        v.x += hres.x; v.y += hres.y; v.z += hres.z; v.w += hres.w;
        x[r] = v;
        float s = (v.x + v.y) + (v.z + v.w);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        sum[r] = s;
    }
    if (g == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) sh_red[w * RT + r * 16 + n] = sum[r];
    }
    __syncthreads();
    float mu[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) s += sh_red[ww * RT + r * 16 + n];
        mu[r] = s * (1.0f / D);
        const float dx = x[r].x - mu[r], dy = x[r].y - mu[r], dz = x[r].z - mu[r], dw = x[r].w - mu[r];
        float q = (dx * dx + dy * dy) + (dz * dz + dw * dw);
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        sum[r] = q;
    }
    float* sh_red2 = sh_red + NW * RT;
    if (g == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) sh_red2[w * RT + r * 16 + n] = sum[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long t = t0 + r * 16 + n;
        float q = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) q += sh_red2[ww * RT + r * 16 + n];
        const float rs = rsqrtf(q * (1.0f / D) + p.eps);
        if (t < p.T) {
            st4(p.h1 + t * D + f0, make_float4((x[r].x - mu[r]) * rs * gam.x + bet.x, (x[r].y - mu[r]) * rs * gam.y + bet.y,
                                               (x[r].z - mu[r]) * rs * gam.z + bet.z, (x[r].w - mu[r]) * rs * gam.w + bet.w));
            if (TRAIN && w == 0 && g == 0) { p.mean[t] = mu[r]; p.rstd[t] = rs; }
        }
    }
}
template <int D, int R, bool HS = false>
__global__ __launch_bounds__(D * 4) void xlnet_oproj_ln_kernel(OProjParams p) {
    extern __shared__ uint16_t smem16[];
    if (p.ao != nullptr) oproj_body<D, R, true, HS>(p, smem16);
    else oproj_body<D, R, false, HS>(p, smem16);
}

// ---------------------------------------------------------------------------------------------- LayerNorm 1 backward + d attn_vec
struct Ln1BwdParams {
    const float *dy, *ao, *h, *mean, *rstd, *gamma;     // d loss / d h1; o-projection output (pre dropout); layer input; stats
    const uint16_t* planes;                             // OT planes [D][D] (bf16 x 3, or fp16 x 2 with *wscale)
    const float* wscale;
    float *dh, *dao, *dav;                              // [T, D] each, overwritten: residual part of d h, d attn_out, d attn_vec
    float* part;                                        // [nWG][2 D] partial sums (d gamma | d beta)
    long T;
    DropCfg drop;
};

template <int D, int R, bool HS = false>
__global__ __launch_bounds__(D * 4) void xlnet_ln1_bwd_kernel(Ln1BwdParams p) {
    constexpr int NW = D / 16, RT = 16 * R, PH = D + 16, PLN = RT * PH;
    extern __shared__ uint16_t smem16[];
    uint16_t* sh_d = smem16;                                         // [3][RT][PH] d attn_out planes
    float* sh_part = reinterpret_cast<float*>(sh_d + 3 * PLN);       // [NW][2][D]
    float* sh_inv = reinterpret_cast<float*>(sh_d + 2 * PLN);        // HS: [RT] inverse row scales
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, g = lane >> 4;
    const long t0 = (long)blockIdx.x * RT;
    AFrag<D> a;
    AFragH<D> ah;
    if constexpr (HS) load_a2h<D>(ah, p.planes + (long)(16 * w + n) * D + 8 * g, (long)D * D);
    else load_a3<D>(a, p.planes + (long)(16 * w + n) * D + 8 * g, (long)D * D);
    {
        const int c0 = lane * 2;
        const bool act_lane = c0 < D;
        float gam[2], pg[2] = {0.f, 0.f}, pb[2] = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 2; ++e) gam[e] = act_lane ? p.gamma[c0 + e] : 0.f;
        // rows RB at a time, every load of the batch in flight before the first reduction (as xlnet_ff_bwd_kernel's phase A)
        constexpr int RB = 5;
        for (int row0 = w; row0 < RT; row0 += NW * RB) {
            float2 fo_[RB], hh_[RB], dd_[RB];
            float mu_[RB], rs_[RB];
#pragma unroll
            for (int b = 0; b < RB; ++b) {          // unconditional loads from clamped addresses
                const long tc = min(t0 + min(row0 + b * NW, RT - 1), p.T - 1);
                mu_[b] = p.mean[tc]; rs_[b] = p.rstd[tc];
                const int cc = act_lane ? c0 : 0;
                fo_[b] = *reinterpret_cast<const float2*>(p.ao + tc * D + cc);
                hh_[b] = *reinterpret_cast<const float2*>(p.h + tc * D + cc);
                dd_[b] = *reinterpret_cast<const float2*>(p.dy + tc * D + cc);
            }
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const int row = row0 + b * NW;
                if (row >= RT) break;           // wave-uniform
            const long t = t0 + row;
            float dxa[2] = {0.f, 0.f};
            if (t < p.T) {      // wave-uniform
                const float mu = mu_[b], rs = rs_[b];
                float xh[2] = {0.f, 0.f}, gg[2] = {0.f, 0.f}, dyv[2] = {0.f, 0.f}, m[2] = {1.f, 1.f}, dx[2];
                float s1 = 0.f, s2 = 0.f;
                if (act_lane) {
                    drop_scale_vec<2>(p.drop, (unsigned long long)t * D + c0, true, m);
                    const float2 fo = fo_[b], hh = hh_[b], dd = dd_[b];
                    const float xv[2] = {fo.x * m[0] + hh.x, fo.y * m[1] + hh.y};
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
                    dx[e] = act_lane ? rs * (gg[e] - s1 - xh[e] * s2) : 0.f;
                    dxa[e] = dx[e] * m[e];
                    pg[e] += dyv[e] * xh[e];
                    pb[e] += dyv[e];
                }
                if (act_lane) {
                    *reinterpret_cast<float2*>(p.dh + t * D + c0) = make_float2(dx[0], dx[1]);
                    *reinterpret_cast<float2*>(p.dao + t * D + c0) = make_float2(dxa[0], dxa[1]);
                }
            }
            if constexpr (HS) {      // the gradient row positioned by its own power-of-two scale (wave-uniform)
                float m = fmaxf(fabsf(dxa[0]), fabsf(dxa[1]));
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
                const float sc = pow2_scale(m);
                if (act_lane) {
                    uint32_t wd[2];
                    cut2h(dxa[0] * sc, dxa[1] * sc, wd);
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) *reinterpret_cast<uint32_t*>(sh_d + pl * PLN + row * PH + c0) = wd[pl];
                }
                if (lane == 0) sh_inv[row] = 1.f / sc;
            } else if (act_lane) {
                uint32_t wd[3];
                cut3(dxa[0], dxa[1], wd);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint32_t*>(sh_d + pl * PLN + row * PH + c0) = wd[pl];
            }
            }
        }
        if (act_lane) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                sh_part[(w * 2 + 0) * D + c0 + e] = pg[e];
                sh_part[(w * 2 + 1) * D + c0 + e] = pb[e];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * D; i += NW * 64) {
        float s = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) s += sh_part[ww * 2 * D + i];
        p.part[(long)blockIdx.x * 2 * D + i] = s;
    }
    // d attn_vec^T[nd][token] = sum_h o[h][nd] d ao[token][h]
    f32x4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = zero4();
    if constexpr (HS) product3h<D, R, PH>(ah, sh_d + n * PH + 8 * g, PLN, acc);
    else product3<D, R, PH>(a, sh_d + n * PH + 8 * g, PLN, acc);
    const float iw = HS ? 1.f / *p.wscale : 1.f;
    const int f0 = 16 * w + 4 * g;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long t = t0 + r * 16 + n;
        const float sc = HS ? sh_inv[r * 16 + n] * iw : 1.f;
        if (t < p.T) st4(p.dav + t * D + f0, make_float4(acc[r][0] * sc, acc[r][1] * sc, acc[r][2] * sc, acc[r][3] * sc));
    }
}

// ---------------------------------------------------------------------------------------------- d h from d q, d k, d v
struct DhParams {
    const float* dqkv;        // [3][T][D]
    const uint16_t* planes;   // QKVN planes [D][3 D] (bf16 x 3, or fp16 x 2 with wscale[0..2])
    const float* wscale;
    float* dh;                // [T, D], ACCUMULATED into (holds the LayerNorm-backward residual part)
    long T;
    DropCfg drop_in;          // p > 0 (first layer): dh is the gradient of the DROPPED model input: masked in the final store
};

template <int D, int R, bool HS = false>
__global__ __launch_bounds__(D * 4) void xlnet_dh_kernel(DhParams p) {
    constexpr int NW = D / 16, NT = NW * 64, RT = 16 * R, PH = D + 16, PLN = RT * PH;
    extern __shared__ uint16_t smem16[];     // [2][3][RT][PH] (HS: two planes + [RT] inverse token scales per buffer)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 15, g = lane >> 4;
    const long t0 = (long)blockIdx.x * RT;
    const long TD = p.T * D;
    const uint16_t* wp = p.planes + (long)(16 * w + n) * 3 * D + 8 * g;
    const long wpl = 3L * D * D;
    f32x4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = zero4();
    auto inv_of = [&](int buf) { return reinterpret_cast<float*>(smem16 + buf * 3 * PLN + 2 * PLN); };
    // HS: a gradient tile is REQUESTED (raw rows into registers) before the product of the tile in front of it and cut into
    // its planes after it: the load latency of d k / d v sits under the matrix instructions of d q / d k (round 5; the
    // one-call tile_to_planes_h in front of each product exposed it three times per launch).  Same arithmetic per element.
    constexpr int G4 = D / 4, TIT = (RT * G4 + NT - 1) / NT;
    float4 raw[TIT];
    auto load_tile = [&](const float* __restrict__ src) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < TIT; ++k) {
            const int i = tid + k * NT;
            const bool live = i < RT * G4;
            const int row = live ? i / G4 : RT - 1, c4 = live ? (i % G4) * 4 : 0;
            raw[k] = ld4(src + min(t0 + row, p.T - 1) * D + c4);
        }
    };
    auto store_tile = [&](uint16_t* sh, float* sh_inv) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < TIT; ++k) {
            const int i = tid + k * NT;
            const bool live = i < RT * G4;
            const int row = live ? i / G4 : RT - 1, c4 = live ? (i % G4) * 4 : 0;
            const float4 v = raw[k];
            float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
#pragma unroll
            for (int o = G4 / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            const float sc = pow2_scale(m);
            if (live) {
                uint32_t w0[2], w1[2];
                cut2h(v.x * sc, v.y * sc, w0);
                cut2h(v.z * sc, v.w * sc, w1);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    *reinterpret_cast<uint2*>(sh + pl * PLN + row * PH + c4) = make_uint2(w0[pl], w1[pl]);
                if (c4 == 0) sh_inv[row] = 1.f / sc;
            }
        }
    };
    if constexpr (HS) { load_tile(p.dqkv); store_tile(smem16, inv_of(0)); }
    else tile_to_planes<D, RT, NT, PH>(p.dqkv, t0, p.T, smem16, tid);
    __syncthreads();
#pragma unroll
    for (int z = 0; z < 3; ++z) {
        uint16_t* cur = smem16 + (z & 1) * 3 * PLN;
        if constexpr (HS) {
            // every gradient tile carries its own token scales and every weight block its own scale: the three products
            // are scaled back one by one before they are added
            AFragH<D> a;
            load_a2h<D>(a, wp + z * D, wpl);
            if (z < 2) load_tile(p.dqkv + (z + 1) * TD);
            f32x4 part[R];
#pragma unroll
            for (int r = 0; r < R; ++r) part[r] = zero4();
            product3h<D, R, PH>(a, cur + n * PH + 8 * g, PLN, part);
            const float iw = 1.f / p.wscale[z];
            const float* inv = inv_of(z & 1);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float sc = inv[r * 16 + n] * iw;
                acc[r][0] += part[r][0] * sc; acc[r][1] += part[r][1] * sc; acc[r][2] += part[r][2] * sc; acc[r][3] += part[r][3] * sc;
            }
            if (z < 2) store_tile(smem16 + ((z + 1) & 1) * 3 * PLN, inv_of((z + 1) & 1));
        } else {
            AFrag<D> a;
            load_a3<D>(a, wp + z * D, wpl);
            if (z < 2) tile_to_planes<D, RT, NT, PH>(p.dqkv + (z + 1) * TD, t0, p.T, smem16 + ((z + 1) & 1) * 3 * PLN, tid);
            product3<D, R, PH>(a, cur + n * PH + 8 * g, PLN, acc);
        }
        if (z < 2) __syncthreads();
    }
    const int k0 = 16 * w + 4 * g;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long t = t0 + r * 16 + n;
        if (t < p.T) {
            const float4 o = ld4(p.dh + t * D + k0);
            float4 v = make_float4(o.x + acc[r][0], o.y + acc[r][1], o.z + acc[r][2], o.w + acc[r][3]);
            if (p.drop_in.p > 0.f) {        // workgroup-uniform
                const float4 m = drop_scale4(p.drop_in, (unsigned long long)t * D + k0);
                v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
            }
            st4(p.dh + t * D + k0, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------- host side
static int pick_r(long T) { return t4r_xlnet_pick_r(T, false); }

#define ATTN_DISPATCH(CALL, D, R)                                                              \
    switch ((D) * 8 + (R)) {                                                                   \
        case 32 * 8 + 1: CALL(32, 1); break;   case 32 * 8 + 2: CALL(32, 2); break;            \
        case 32 * 8 + 3: CALL(32, 3); break;   case 32 * 8 + 5: CALL(32, 5); break;            \
        case 64 * 8 + 1: CALL(64, 1); break;   case 64 * 8 + 2: CALL(64, 2); break;            \
        case 64 * 8 + 3: CALL(64, 3); break;   case 64 * 8 + 5: CALL(64, 5); break;            \
        case 128 * 8 + 1: CALL(128, 1); break; case 128 * 8 + 2: CALL(128, 2); break;          \
        case 128 * 8 + 3: CALL(128, 3); break; case 128 * 8 + 5: CALL(128, 5); break;          \
        default: t4r_set_error("xlnet fused: no instantiation"); return -1;                    \
    }


// q, k, v = h @ W_{q,k,v}: out = qkv [3][T][D] (one launch).  planes: t4r_xlnet_layer_prepare's buffer.
extern "C" int t4r_xlnet_qkv_proj(void* stream, const float* h, const float* planes, float* qkv, long T, int D) {
    if (T <= 0) return 0;
    T4R_CHECK_ARG(t4r_xlnet_fused_supported(D) && h && planes && qkv, "xlnet_qkv_proj: bad arguments");
    const int R = pick_r(T);
    const bool hs = t4r_xlnet_body_fp16x2();
    const LayerPlanesH PH_ = carve_planes_h(planes, D);
    const uint16_t* wq = hs ? PH_.QKVT : carve_planes(planes, D).QKVT;
    ProjParams p{h, {wq, wq + (long)D * D, wq + 2L * D * D, nullptr}, 3L * D * D, {qkv, qkv + T * D, qkv + 2 * T * D, nullptr}, T,
                 {PH_.scale + HS_Q, PH_.scale + HS_K, PH_.scale + HS_V, nullptr}};
    hipStream_t st = (hipStream_t)stream;
#define CALL(DD, RR)                                                                                             \
    {                                                                                                            \
        const size_t smem = (size_t)3 * 16 * RR * (DD + 16) * 2;       /* HS: two planes + the token scales fit the same size */ \
        if (hs) {                                                                                                \
            { static T4rLdsAttr once; t4r_ensure_dynamic_lds((const void*)xlnet_proj_kernel<DD, RR, 3, true>, smem, once); } \
            hipLaunchKernelGGL((xlnet_proj_kernel<DD, RR, 3, true>), dim3((unsigned)((T + 16 * RR - 1) / (16 * RR))), dim3(DD * 4), smem, st, p); \
        } else {                                                                                                 \
            { static T4rLdsAttr once; t4r_ensure_dynamic_lds((const void*)xlnet_proj_kernel<DD, RR, 3>, smem, once); } \
            hipLaunchKernelGGL((xlnet_proj_kernel<DD, RR, 3>), dim3((unsigned)((T + 16 * RR - 1) / (16 * RR))), dim3(DD * 4), smem, st, p); \
        }                                                                                                        \
    }
    ATTN_DISPATCH(CALL, D, R)
#undef CALL
    T4R_LAUNCH_CHECK();
    return 0;
}

// k_r = pos @ r: pos [rows, D] (the positional encoding [2L, D], or its per-session dropped copy [B 2L, D]) -> kr [rows, D]
extern "C" int t4r_xlnet_kr_proj(void* stream, const float* pos, const float* planes, float* kr, long rows, int D) {
    if (rows <= 0) return 0;
    T4R_CHECK_ARG(t4r_xlnet_fused_supported(D) && pos && planes && kr, "xlnet_kr_proj: bad arguments");
    const int R = pick_r(rows);
    const bool hs = t4r_xlnet_body_fp16x2();
    const LayerPlanesH PH_ = carve_planes_h(planes, D);
    ProjParams p{pos, {hs ? PH_.RT : carve_planes(planes, D).RT, nullptr, nullptr, nullptr}, (long)D * D, {kr, nullptr, nullptr, nullptr}, rows,
                 {PH_.scale + HS_R, nullptr, nullptr, nullptr}};
    hipStream_t st = (hipStream_t)stream;
#define CALL(DD, RR)                                                                                             \
    {                                                                                                            \
        const size_t smem = (size_t)3 * 16 * RR * (DD + 16) * 2;                                                 \
        if (hs) {                                                                                                \
            { static T4rLdsAttr once; t4r_ensure_dynamic_lds((const void*)xlnet_proj_kernel<DD, RR, 1, true>, smem, once); } \
            hipLaunchKernelGGL((xlnet_proj_kernel<DD, RR, 1, true>), dim3((unsigned)((rows + 16 * RR - 1) / (16 * RR))), dim3(DD * 4), smem, st, p); \
        } else {                                                                                                 \
            { static T4rLdsAttr once; t4r_ensure_dynamic_lds((const void*)xlnet_proj_kernel<DD, RR, 1>, smem, once); } \
            hipLaunchKernelGGL((xlnet_proj_kernel<DD, RR, 1>), dim3((unsigned)((rows + 16 * RR - 1) / (16 * RR))), dim3(DD * 4), smem, st, p); \
        }                                                                                                        \
    }
    ATTN_DISPATCH(CALL, D, R)
#undef CALL
    T4R_LAUNCH_CHECK();
    return 0;
}

// The prologue of a whole layer stack in two launches instead of two per layer: every layer's weight planes (they do not
// change inside a step) and every layer's positional keys k_r = pos @ r_l (the positional encoding -- with its dropout
// mask, drawn once per forward -- is the same for all layers, so its tile is read and cut once for up to four layers).
// planes[l] / kr[l]: the layer's own buffers (inside its t4r_xlnet_layer_fwd workspace: t4r_xlnet_layer_ws_offsets).
// The next t4r_xlnet_stack_prepare of this thread MAKES the dropped positional rows itself (round 6): `pos` is then the plain
// [period, D] positional encoding, pos_rows = B x period, and the projection kernel masks row t = pos[t % period] with the
// pos_emb dropout of this forward (key (seed, ctr)) and writes the dropped rows to `out` [pos_rows, D] (what the caller used to
// produce with t4r_dropout before the call, and what the backward's d r contraction reads).  Consumed by that call.
static thread_local GenRows g_stack_gen = {nullptr, 0, DropCfg(), nullptr};
extern "C" void t4r_xlnet_stack_pos_dropout(float p, unsigned long long seed, unsigned long long ctr, long period, float* out) {
    g_stack_gen.base = nullptr; g_stack_gen.period = period; g_stack_gen.drop = make_drop(p, seed, ctr); g_stack_gen.out = out;
}
extern "C" int t4r_xlnet_stack_prepare(void* stream, const float* const* params_all, int n_layers, int D,
                                       float* const* planes, const float* pos, long pos_rows, float* const* kr) {
    GenRows gen = g_stack_gen;
    g_stack_gen.out = nullptr;
    T4R_CHECK_ARG(t4r_xlnet_fused_supported(D) && params_all && planes && n_layers >= 1, "xlnet_stack_prepare: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    for (int l0 = 0; l0 < n_layers; l0 += 4) {
        const int nl = min(4, n_layers - l0);
        PlaneJobs js;
        AmaxJobs aj;
        js.n = 0; js.total = 0; aj.n = 0;
        for (int l = l0; l < l0 + nl; ++l) {
            const float* const* pr = params_all + (long)l * 15;
            T4R_CHECK_ARG(planes[l] && pr[0] && pr[1] && pr[2] && pr[3] && pr[4] && pr[9] && pr[10] && pr[11],
                          "xlnet_stack_prepare: null pointer");
            add_layer_jobs(js, aj, pr[0], pr[1], pr[2], pr[3], pr[4], pr[9], pr[10], pr[11], D, planes[l]);
        }
        if (launch_jobs(st, js, aj)) return -1;
        if (!pos || pos_rows <= 0) continue;
        T4R_CHECK_ARG(kr, "xlnet_stack_prepare: null k_r pointers");
        const bool hs = t4r_xlnet_body_fp16x2();
        ProjParams p{pos, {nullptr, nullptr, nullptr, nullptr}, (long)D * D, {nullptr, nullptr, nullptr, nullptr}, pos_rows,
                     {nullptr, nullptr, nullptr, nullptr}};
        if (gen.out) {          // generated rows: needs the two-way fp16 form (the only one the product build runs)
            T4R_CHECK_ARG(hs && gen.period > 0 && pos_rows % gen.period == 0 && gen.drop.p > 0.f,
                          "xlnet_stack_prepare: the fused pos_emb dropout needs the fp16 form, p > 0 and pos_rows a multiple of the period");
            p.gen = gen;
            p.gen.base = pos;
        }
        for (int m = 0; m < nl; ++m) {
            const LayerPlanesH PH_ = carve_planes_h(planes[l0 + m], D);
            p.w[m] = hs ? PH_.RT : carve_planes(planes[l0 + m], D).RT;
            p.wscale[m] = PH_.scale + HS_R;
            p.out[m] = kr[l0 + m];
        }
        const int R = pick_r(pos_rows);
#define CALLN(DD, RR, NMV)                                                                                       \
    {                                                                                                            \
        const size_t smem = (size_t)3 * 16 * RR * (DD + 16) * 2;                                                 \
        if (hs) {                                                                                                \
            { static T4rLdsAttr once; t4r_ensure_dynamic_lds((const void*)xlnet_proj_kernel<DD, RR, NMV, true>, smem, once); } \
            hipLaunchKernelGGL((xlnet_proj_kernel<DD, RR, NMV, true>), dim3((unsigned)((pos_rows + 16 * RR - 1) / (16 * RR))), dim3(DD * 4), smem, st, p); \
        } else {                                                                                                 \
            { static T4rLdsAttr once; t4r_ensure_dynamic_lds((const void*)xlnet_proj_kernel<DD, RR, NMV>, smem, once); } \
            hipLaunchKernelGGL((xlnet_proj_kernel<DD, RR, NMV>), dim3((unsigned)((pos_rows + 16 * RR - 1) / (16 * RR))), dim3(DD * 4), smem, st, p); \
        }                                                                                                        \
    }
#define CALL1(DD, RR) CALLN(DD, RR, 1)
#define CALL2(DD, RR) CALLN(DD, RR, 2)
#define CALL3(DD, RR) CALLN(DD, RR, 3)
#define CALL4(DD, RR) CALLN(DD, RR, 4)
        if (nl == 1) { ATTN_DISPATCH(CALL1, D, R) } else if (nl == 2) { ATTN_DISPATCH(CALL2, D, R) }
        else if (nl == 3) { ATTN_DISPATCH(CALL3, D, R) } else { ATTN_DISPATCH(CALL4, D, R) }
#undef CALL1
#undef CALL2
#undef CALL3
#undef CALL4
#undef CALLN
        T4R_LAUNCH_CHECK();
    }
    return 0;
}

// h1 = LayerNorm(dropout(attn_vec @ o^T) + h).  Training: ao [T, D] (pre-dropout projection), mean, rstd [T] saved;
// inference: all three NULL and drop_p = 0.
extern "C" int t4r_xlnet_oproj_ln(void* stream, const float* av, const float* h, const float* planes, const float* gamma,
                                  const float* beta, float* ao, float* mean, float* rstd, float* h1, long T, int D, float eps,
                                  float drop_p, unsigned long long seed, unsigned long long ctr_hi) {
    if (T <= 0) return 0;
    T4R_CHECK_ARG(t4r_xlnet_fused_supported(D) && av && h && planes && gamma && beta && h1, "xlnet_oproj_ln: bad arguments");
    const bool train = ao != nullptr;
    T4R_CHECK_ARG((mean != nullptr) == train && (rstd != nullptr) == train, "xlnet_oproj_ln: ao, mean, rstd go together");
    T4R_CHECK_ARG(train || drop_p == 0.f, "xlnet_oproj_ln: dropout needs the saved activations");
    const int R = pick_r(T);
    const bool hs = t4r_xlnet_body_fp16x2();
    const LayerPlanesH PH_ = carve_planes_h(planes, D);
    OProjParams p{av, h, gamma, beta, hs ? PH_.ON : carve_planes(planes, D).ON, PH_.scale + HS_O, ao, mean, rstd, h1, T, eps,
                  make_drop(drop_p, seed, ctr_hi)};
    hipStream_t st = (hipStream_t)stream;
#define CALL(DD, RR)                                                                                             \
    {                                                                                                            \
        const size_t smem = (size_t)3 * 16 * RR * (DD + 16) * 2 + (size_t)2 * (DD / 16) * 16 * RR * 4;           \
        const dim3 grid((unsigned)((T + 16 * RR - 1) / (16 * RR)));                                              \
        if (hs) {                                                                                                \
            { static T4rLdsAttr once; t4r_ensure_dynamic_lds((const void*)xlnet_oproj_ln_kernel<DD, RR, true>, smem, once); } \
            hipLaunchKernelGGL((xlnet_oproj_ln_kernel<DD, RR, true>), grid, dim3(DD * 4), smem, st, p);          \
        } else {                                                                                                 \
            { static T4rLdsAttr once; t4r_ensure_dynamic_lds((const void*)xlnet_oproj_ln_kernel<DD, RR>, smem, once); } \
            hipLaunchKernelGGL((xlnet_oproj_ln_kernel<DD, RR>), grid, dim3(DD * 4), smem, st, p);                \
        }                                                                                                        \
    }
    ATTN_DISPATCH(CALL, D, R)
#undef CALL
    T4R_LAUNCH_CHECK();
    return 0;
}

extern "C" long t4r_xlnet_ln1_bwd_part_floats(long T, int D) { return ((T + 15) / 16) * 2L * D; }

// LayerNorm-1 backward + d attn_vec.  dy = d loss / d h1.  Overwrites dh (residual part of d loss / d h), dao (d loss / d of
// the o-projection output: rows of the o weight gradient d o += dao^T @ attn_vec) and dav (d loss / d attn_vec);
// d_gamma, d_beta ACCUMULATED.  part: t4r_xlnet_ln1_bwd_part_floats(T, D) floats.
extern "C" int t4r_xlnet_ln1_bwd(void* stream, const float* dy, const float* ao, const float* h, const float* mean,
                                 const float* rstd, const float* gamma, const float* planes, float* dh, float* dao, float* dav,
                                 float* d_gamma, float* d_beta, float* part, long T, int D, float drop_p,
                                 unsigned long long seed, unsigned long long ctr_hi) {
    if (T <= 0) return 0;
    T4R_CHECK_ARG(t4r_xlnet_fused_supported(D) && dy && ao && h && mean && rstd && gamma && planes && dh && dao && dav && part,
                  "xlnet_ln1_bwd: bad arguments");
    const int R = t4r_xlnet_pick_r(T, true);
    const int nwg = (int)((T + 16 * R - 1) / (16 * R));
    const bool hs = t4r_xlnet_body_fp16x2();
    const LayerPlanesH PH_ = carve_planes_h(planes, D);
    Ln1BwdParams p{dy, ao, h, mean, rstd, gamma, hs ? PH_.OT : carve_planes(planes, D).OT, PH_.scale + HS_O, dh, dao, dav, part, T,
                   make_drop(drop_p, seed, ctr_hi)};
    hipStream_t st = (hipStream_t)stream;
#define CALL(DD, RR)                                                                                             \
    {                                                                                                            \
        const size_t smem = (size_t)3 * 16 * RR * (DD + 16) * 2 + (size_t)(DD / 16) * 2 * DD * 4;                \
        if (hs) {                                                                                                \
            { static T4rLdsAttr once; t4r_ensure_dynamic_lds((const void*)xlnet_ln1_bwd_kernel<DD, RR, true>, smem, once); } \
            hipLaunchKernelGGL((xlnet_ln1_bwd_kernel<DD, RR, true>), dim3((unsigned)nwg), dim3(DD * 4), smem, st, p); \
        } else {                                                                                                 \
            { static T4rLdsAttr once; t4r_ensure_dynamic_lds((const void*)xlnet_ln1_bwd_kernel<DD, RR>, smem, once); } \
            hipLaunchKernelGGL((xlnet_ln1_bwd_kernel<DD, RR>), dim3((unsigned)nwg), dim3(DD * 4), smem, st, p);  \
        }                                                                                                        \
    }
    ATTN_DISPATCH(CALL, D, R)
#undef CALL
    T4R_LAUNCH_CHECK();
    return t4r_reduce_partials_launch(st, part, nwg, d_gamma, D, 1, d_beta, D, 1, nullptr, 0, 0);
}

// dh [T, D] += d q @ W_q^T + d k @ W_k^T + d v @ W_v^T   (dqkv [3][T][D])
// set and cleared inside one layer call (T4R_LAYER_FUSE_INPUT): the next t4r_xlnet_dh masks its result with the input dropout
static thread_local int g_dh_in_on = 0;
static thread_local float g_dh_in_p = 0.f;
static thread_local unsigned long long g_dh_in_seed = 0, g_dh_in_ctr = 0;
void t4r_xlnet_dh_input_dropout(int on, float p, unsigned long long seed, unsigned long long ctr) {
    g_dh_in_on = on; g_dh_in_p = p; g_dh_in_seed = seed; g_dh_in_ctr = ctr;
}
extern "C" int t4r_xlnet_dh(void* stream, const float* dqkv, const float* planes, float* dh, long T, int D) {
    if (T <= 0) return 0;
    T4R_CHECK_ARG(t4r_xlnet_fused_supported(D) && dqkv && planes && dh, "xlnet_dh: bad arguments");
    const int R = t4r_xlnet_pick_r(T, true);
    const bool hs = t4r_xlnet_body_fp16x2();
    const LayerPlanesH PH_ = carve_planes_h(planes, D);
    DhParams p{dqkv, hs ? PH_.QKVN : carve_planes(planes, D).QKVN, PH_.scale + HS_Q, dh, T,
               make_drop(g_dh_in_on ? g_dh_in_p : 0.f, g_dh_in_seed, g_dh_in_ctr)};
    hipStream_t st = (hipStream_t)stream;
#define CALL(DD, RR)                                                                                             \
    {                                                                                                            \
        const size_t smem = (size_t)2 * 3 * 16 * RR * (DD + 16) * 2;                                             \
        const dim3 grid((unsigned)((T + 16 * RR - 1) / (16 * RR)));                                              \
        if (hs) {                                                                                                \
            { static T4rLdsAttr once; t4r_ensure_dynamic_lds((const void*)xlnet_dh_kernel<DD, RR, true>, smem, once); } \
            hipLaunchKernelGGL((xlnet_dh_kernel<DD, RR, true>), grid, dim3(DD * 4), smem, st, p);                \
        } else {                                                                                                 \
            { static T4rLdsAttr once; t4r_ensure_dynamic_lds((const void*)xlnet_dh_kernel<DD, RR>, smem, once); }   \
            hipLaunchKernelGGL((xlnet_dh_kernel<DD, RR>), grid, dim3(DD * 4), smem, st, p);                      \
        }                                                                                                        \
    }
    ATTN_DISPATCH(CALL, D, R)
#undef CALL
    T4R_LAUNCH_CHECK();
    return 0;
}
