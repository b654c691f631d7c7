// One XLNet layer (relative attention + feed-forward, both post-LN), forward and backward, as
// a fixed chain of launches on one HIP stream.  Restates HF modeling_xlnet.py
//   XLNetRelativeAttention.forward (g=None branch) :245-282   + post_attention :142-152
//   XLNetFeedForward.forward                        :297-305
//   XLNetLayer.forward                              :308-353
// with the hyper-parameters fixed by XLNetConfig.build (transformers4rec/config/transformer.py:
// 432-482): d_inner = 4 d_model, gelu(erf), layer_norm_eps as passed (0.03), attn_type "bi",
// no masks, no mems.  Dropout = identity here (eval / p = 0).
//
// Layout (all fp32, row-major, token t = b*L + l):
//   h [T,D] -> q,k,v = h @ {q,k,v}[D, n*dh]           (one batched MFMA GEMM when q,k,v are adjacent)
//   k_r = pos_emb[2L,D] @ r[D, n*dh]                  (once per layer, NOT per batch row)
//   attn_vec [T,D] = rel-attention(q,k,v,k_r)         (xlnet_attn.hip)
//   h1 = LN(attn_vec @ o^T + h) ; h2 = LN(W2 gelu(W1 h1 + b1) + b2 + h1)
// Saved-for-backward activations live in one caller-provided workspace (offsets below).
#include "t4r_common.h"
#include <stdlib.h>

int t4r_gemm_launch(hipStream_t stream, int transA, int transB, int M, int N, int K, float alpha,
                    const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                    const float* bias, int epilogue, float* aux, long ldaux, int splitk,
                    int accumulate, int batch, long sA, long sB, long sC, const DropCfg* drop);
extern "C" {
int t4r_add_layernorm_fwd(void*, const float*, const float*, const float*, const float*, float*,
                          float*, float*, int, int, float, float, unsigned long long, unsigned long long);
int t4r_add_layernorm_bwd(void*, const float*, const float*, const float*, const float*,
                          const float*, const float*, float*, float*, float*, float*, float*, int, int,
                          int, float, unsigned long long, unsigned long long);
int t4r_act_bwd_bias(void*, const float*, const float*, float*, float*, float*, long, int, int, float,
                     unsigned long long, unsigned long long);
int t4r_colsum(void*, const float*, float*, float*, long, int, long);
long t4r_colreduce_ws_floats(long, int);
}
void t4r_reduce_redirect(hipStream_t side, hipEvent_t* events, int n_events);   // elementwise.hip
void t4r_gemm_operand_amax(const float* a, const float* b, int n);               // gemm_f32.hip: next launch in the fp16 split form
void t4r_xlnet_ff_amax_buffers(float* h1, float* act, float* dpre, float* dfo);  // xlnet_fused.hip
void t4r_xlnet_ff_final_dropout(int on, unsigned long long ctr);                 // xlnet_fused.hip
// layer_idx carries flags above its low byte (the Philox counters use the low byte only, ctr_hi):
//   T4R_LAYER_FUSE_FINAL  this is the LAST layer of the stack and the model runs in training mode: the model-level output
//                         dropout (HF modeling_xlnet.py:1177, key (offset, 255, SITE_FINAL)) is applied by this layer's
//                         feed-forward kernels -- to h_out in the forward, to dh_out on load in the backward -- instead of by
//                         two element-wise launches over [T, D] around the stack.  Needs the fused kernels (d_model 32/64/128).
#define T4R_LAYER_FUSE_FINAL 0x100
//   T4R_LAYER_FUSE_INPUT  this is the FIRST layer of the stack and `h` is the model's UNDROPPED input: the model-level input
//                         dropout (HF :1116, key (offset, 255, SITE_INPUT)) is applied by the attention-block kernel on load
//                         (the dropped rows are kept in the layer's workspace for the backward), and the backward masks d h
//                         in xlnet_dh's final store -- instead of two element-wise launches over [T, D].  Needs the one-kernel
//                         attention forward (t4r_xlnet_attn_block_supported) and the fused kernels.
#define T4R_LAYER_FUSE_INPUT 0x200
void t4r_xlnet_attn_block_input_dropout(int on, unsigned long long ctr, float* hin);     // xlnet_attn_block.hip
void t4r_xlnet_dh_input_dropout(int on, float p, unsigned long long seed, unsigned long long ctr);   // xlnet_fused_attn.hip
int t4r_xlnet_ff_amax_count(long T);
int t4r_xlnet_ff_amax_count_bwd(long T);
void t4r_gemm_operand_amax2(const float* a, int na, const float* b, int nb);   // gemm_f32.hip: the two producers ran different grids
long t4r_xlnet_ff_amax_slots(long T);
bool t4r_xlnet_body_fp16x2();
void t4r_splitk_sink_begin(float* ws, long cap_floats);                          // gemm_f32.hip: deterministic split-K
int t4r_splitk_sink_flush(hipStream_t st);
void t4r_splitk_sink_end();
extern "C" {
int t4r_dropout(void*, const float*, float*, unsigned char*, long, long, float, unsigned long long,
                unsigned long long);
int t4r_xlnet_attn_fwd(void*, const float*, const float*, const float*, const float*, const float*,
                       const float*, float*, float*, int, int, int, int, int, float, unsigned long long,
                       unsigned long long, const int*);
int t4r_xlnet_attn_bwd(void*, const float*, const float*, const float*, const float*, const float*,
                       const float*, const float*, const float*, const float*, float*, float*, float*,
                       float*, float*, float*, float*, int, int, int, int, int, float,
                       unsigned long long, unsigned long long, const int*);
long t4r_xlnet_attn_bwd_ws_floats(int, int, int, int);
// xlnet_fused.hip: token-tile-stationary feed-forward block (one launch forward, one backward)
int t4r_xlnet_fused_supported(int D);
long t4r_xlnet_ff_bwd_part_floats(long T, int D);
long t4r_xlnet_layer_planes_floats(int D);
int t4r_xlnet_layer_prepare(void*, const float* const*, int, float*);
int t4r_xlnet_qkv_proj(void*, const float*, const float*, float*, long, int);
int t4r_xlnet_kr_proj(void*, const float*, const float*, float*, long, int);
int t4r_xlnet_oproj_ln(void*, const float*, const float*, const float*, const float*, const float*, float*, float*, float*,
                       float*, long, int, float, float, unsigned long long, unsigned long long);
long t4r_xlnet_ln1_bwd_part_floats(long, int);
int t4r_xlnet_ln1_bwd(void*, const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                      float*, float*, float*, float*, float*, float*, long, int, float, unsigned long long, unsigned long long);
int t4r_xlnet_dh(void*, const float*, const float*, float*, long, int);
// xlnet_attn_block.hip: the attention half as one kernel per direction, exact fp32 matrix instructions
int t4r_xlnet_attn_block_supported(int L, int D, int n_head);
int t4r_xlnet_attn_block_fwd(void*, const float*, const float*, const float*, const float*, long, const float*, const float*,
                             const float*, const float*, float*, float*, float*, float*, float*, float*, float*, int, int, int,
                             int, float, float, unsigned long long, unsigned long long, unsigned long long, const int*);
#ifdef T4R_EXPERIMENTAL     /* tools/experimental: the one-kernel backward of the attention half (measured slower, not in the product library) */
long t4r_xlnet_attn_block_bwd_part_floats(int B, int L, int D, int n_head);
int t4r_xlnet_attn_block_bwd(void*, const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                             const float*, const float*, const float*, const float*, const float*, long, const float*, const float*,
                             const float*, float*, float*, float*, float*, float*, float*, float*, float*, float*, int, int, int, int,
                             float, unsigned long long, unsigned long long, unsigned long long, const int*);
#endif
int t4r_xlnet_ff_fwd(void*, const float*, const float*, const float*, const float*, const float*, const float*,
                     float*, float*, float*, float*, float*, float*, int, int, float, float,
                     unsigned long long, unsigned long long, unsigned long long);
int t4r_xlnet_ff_bwd(void*, const float*, const float*, const float*, const float*, const float*, const float*,
                     const float*, const float*, float*, float*, float*, float*, float*, float*, float*,
                     float*, int, int, float, unsigned long long, unsigned long long, unsigned long long);
}
// T4R_XLNET_FUSED=0 restores the launch chain (A/B timing); default: fused kernels where they exist (d_model 32/64/128)
static bool use_fused(int D) {
    static const int on = [] { const char* e = t4r_exp_getenv("T4R_XLNET_FUSED"); return e ? atoi(e) : 1; }();
    return on && t4r_xlnet_fused_supported(D);
}

// The attention half of the FORWARD as one kernel (csrc/xlnet_attn_block.hip), default ON
// (T4R_XLNET_ATTN_BLOCK=0 restores projection -> core -> o-projection + LayerNorm).  The one-kernel BACKWARD was built,
// tested and measured slower (120-130 us per launch against 118 us for the three launches it would replace: docs/DESIGN_rounds_1_to_4.md,
// round 4); it lives in tools/experimental/ and is compiled only into the A/B variant library (-DT4R_EXPERIMENTAL).
static bool use_attn_block(int L, int D, int n_head) {
    static const int on = [] { const char* e = t4r_exp_getenv("T4R_XLNET_ATTN_BLOCK"); return e ? atoi(e) : 1; }();
    return on && use_fused(D) && t4r_xlnet_attn_block_supported(L, D, n_head);
}
#ifdef T4R_EXPERIMENTAL
static bool use_attn_block_bwd(int L, int D, int n_head) {
    static const int on = [] { const char* e = t4r_exp_getenv("T4R_XLNET_ATTN_BLOCK_BWD"); return e ? atoi(e) : 0; }();
    return on && use_fused(D) && t4r_xlnet_attn_block_supported(L, D, n_head);
}
static long attn_block_bwd_part(int B, int L, int D, int n_head) { return t4r_xlnet_attn_block_bwd_part_floats(B, L, D, n_head); }
#else
static long attn_block_bwd_part(int, int, int, int) { return 0; }
#endif

// dropout sites of one layer (HF modeling_xlnet.py): pos_emb :1143 (model level, but the mask is per
// batch row so k_r becomes per-session), attention probabilities :132, attention output :147,
// FF activation :301, FF output :303.  ctr_hi = (offset << 16) | (layer << 8) | site.
enum { SITE_INPUT = 0, SITE_POS = 1, SITE_PROB = 2, SITE_ATTN_OUT = 3, SITE_FF_ACT = 4, SITE_FF_OUT = 5,
       SITE_FINAL = 6 };
static unsigned long long ctr_hi(unsigned long long offset, int layer, int site) {
    return (offset << 16) | ((unsigned long long)(layer & 0xff) << 8) | (unsigned long long)site;
}
extern "C" unsigned long long t4r_dropout_ctr_hi(unsigned long long offset, int layer, int site) {
    return ctr_hi(offset, layer, site);
}

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_GELU = 2, EPI_BIAS_RELU = 3 };

// parameter order of the `params` / `grads` pointer arrays
enum { P_Q = 0, P_K, P_V, P_O, P_R, P_RWB, P_RRB, P_LN1W, P_LN1B, P_W1, P_B1, P_W2, P_B2, P_LN2W,
       P_LN2B, P_COUNT };

struct LayerWs {
    float *qkv, *kr, *av, *lse, *ao, *mean1, *rstd1, *h1, *ffpre, *ffact, *ffout, *mean2, *rstd2, *pe_b, *planes, *amax, *hin;
    long total;
};

static long align4(long x) { return (x + 3) & ~3L; }

static LayerWs carve(float* base, int B, int L, int D, int n, int per_batch_kr) {
    const long T = (long)B * L;
    LayerWs w;
    long o = 0;
    auto take = [&](long nfl) { float* p = base ? base + o : nullptr; o += align4(nfl); return p; };
    w.qkv = take(3 * T * D);
    w.kr = take((per_batch_kr ? (long)B : 1L) * 2L * L * D);
    w.av = take(T * D);
    w.lse = take((long)B * n * L);
    w.ao = take(T * D);
    w.mean1 = take(T);
    w.rstd1 = take(T);
    w.h1 = take(T * D);
    w.ffpre = take(T * 4 * D);
    w.ffact = take(T * 4 * D);
    w.ffout = take(T * D);
    w.mean2 = take(T);
    w.rstd2 = take(T);
    // dropout(pos_emb) per session, kept for the backward's d r contraction (21 MB per layer at C2:
    // with 288 GB of HBM saving beats regenerating it -- one launch less per layer)
    w.pe_b = per_batch_kr ? take((long)B * 2L * L * D) : nullptr;
    // bf16 planes of the feed-forward weights (fused kernels: cut once in the forward, reused by the backward)
    w.planes = t4r_xlnet_fused_supported(D) ? take(t4r_xlnet_layer_planes_floats(D)) : nullptr;
    // per-workgroup operand maxima of the feed-forward weight gradients (fused kernels -> fp16-split GEMMs): 4 arrays
    w.amax = t4r_xlnet_fused_supported(D) ? take(4 * t4r_xlnet_ff_amax_slots(T)) : nullptr;
    // the dropped input rows of a FIRST layer that applies the model's input dropout itself (T4R_LAYER_FUSE_INPUT); last, so
    // that every other offset is what it was
    w.hin = (per_batch_kr && t4r_xlnet_fused_supported(D)) ? take(T * D) : nullptr;
    w.total = o;
    return w;
}

// dropout != 0: positional keys are per session (k_r [B,2L,D])
extern "C" long t4r_xlnet_layer_ws_floats(int B, int L, int D, int n_head, int dropout) {
    return carve(nullptr, B, L, D, n_head, dropout).total;
}
// where the weight planes and the positional keys k_r live inside a layer's workspace (float offsets; planes -1 when the
// width has no fused kernels): the buffers t4r_xlnet_stack_prepare fills for all layers of a stack at once
extern "C" int t4r_xlnet_layer_ws_offsets(int B, int L, int D, int n_head, int dropout, long* planes_off, long* kr_off) {
    T4R_CHECK_ARG(planes_off && kr_off, "xlnet_layer_ws_offsets: null pointer");
    float* const base = reinterpret_cast<float*>(sizeof(float) * 4);        // any non-null base: only differences are used
    const LayerWs w = carve(base, B, L, D, n_head, dropout);
    *planes_off = w.planes ? (long)(w.planes - base) : -1;
    *kr_off = (long)(w.kr - base);
    return 0;
}
// t4r_xlnet_stack_prepared(1): the following t4r_xlnet_layer_fwd calls of this thread find their weight planes and k_r
// already in their workspace (t4r_xlnet_stack_prepare) and skip both launches (fused path only)
static thread_local int g_stack_prepared = 0;
extern "C" void t4r_xlnet_stack_prepared(int on) { g_stack_prepared = on ? 1 : 0; }
// partial tiles of the layer's split-K weight gradients (gemm_f32.hip: deterministic split-K): the split rule gives at
// most K / 320 + 1 splits per product (>= 20 k-tiles of 16 each), K = T (2 T for d r with per-session k_r); outputs
// 2 x 4 D^2 (FF) + 4 D^2 (q, k, v, o) + D^2 (r)
static long splitk_sink_floats(long T, int D) { return (T / 320 + 2) * 12L * D * D + (2 * T / 320 + 2) * (long)D * D; }
// backward scratch: dqkv [3,T,D] + dav [T,D] + dx [T,D] + dff [T,4D] + dkr [2L,D] + attention partials
extern "C" long t4r_xlnet_layer_bwd_ws_floats(int B, int L, int D, int n_head, int dropout) {
    const long T = (long)B * L;
    const long nkr = (dropout ? (long)B : 1L) * 2L * L * D;
    return align4(3 * T * D) + align4(T * D) + align4(T * D) + align4(T * 4 * D) + align4(nkr) +
           align4(t4r_xlnet_attn_bwd_ws_floats(B, L, D, n_head)) + align4(t4r_colreduce_ws_floats(T, 4 * D)) +
           2 * align4(t4r_colreduce_ws_floats(T, 2 * D)) + align4(t4r_colreduce_ws_floats(T, D)) +
           (dropout ? align4(T * D) : 0) + 2 * align4(T * D) + align4(t4r_xlnet_ff_bwd_part_floats(T, D)) +
           align4(t4r_xlnet_ln1_bwd_part_floats(T, D)) + align4(attn_block_bwd_part(B, L, D, n_head)) +
           align4(splitk_sink_floats(T, D));
}

#define RUN(call)                \
    do {                         \
        int rc__ = (call);       \
        if (rc__ != 0) return rc__; \
    } while (0)

extern "C" int t4r_xlnet_layer_fwd(void* stream, const float* h, const float* pos_emb,
                                   const float* const* params, float* ws, float* h_out, int B, int L,
                                   int D, int n_head, float ln_eps, float drop_p,
                                   unsigned long long seed, unsigned long long offset, int layer_idx,
                                   const int* key_len, const float* pos_emb_b) {
    if (B == 0) return 0;
    T4R_CHECK_ARG(D % n_head == 0 && D % 4 == 0, "xlnet_layer: d_model must divide by n_head and 4");
    T4R_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "xlnet_layer: dropout p in [0, 1)");
    hipStream_t st = (hipStream_t)stream;
    const int T = B * L, dh = D / n_head;
    const int drop = drop_p > 0.f;
    LayerWs w = carve(ws, B, L, D, n_head, drop);
    const float* q_w = params[P_Q];
    const float* k_w = params[P_K];
    const float* v_w = params[P_V];
    const long TD = (long)T * D, DD = (long)D * D;
    auto C = [&](int site) { return ctr_hi(offset, layer_idx, site); };
    const bool fuse_final = (layer_idx & T4R_LAYER_FUSE_FINAL) != 0;
    T4R_CHECK_ARG(!fuse_final || use_fused(D), "xlnet_layer: the fused output dropout needs the fused layer kernels (d_model 32 / 64 / 128)");
    const bool fuse_in = (layer_idx & T4R_LAYER_FUSE_INPUT) != 0 && drop;
    T4R_CHECK_ARG(!fuse_in || (use_fused(D) && use_attn_block(L, D, n_head)),
                  "xlnet_layer: the fused input dropout needs the one-kernel attention forward (L <= 32, d_model 32 / 64 / 128)");
    if (use_fused(D)) {
        // ONE launch cuts the layer's nine weight matrices into bf16 planes; q, k, v in one token-tile launch; k_r; the
        // attention core; o-projection + dropout + residual + LayerNorm in one launch; the feed-forward block in one
        const bool block = use_attn_block(L, D, n_head);
        if (!g_stack_prepared) RUN(t4r_xlnet_layer_prepare(stream, params, D, w.planes));
        if (!block) RUN(t4r_xlnet_qkv_proj(stream, h, w.planes, w.qkv, T, D));
        if (g_stack_prepared) {
            // k_r of every layer came from one launch of the stack prologue
        } else if (drop) {
            const float* pe_b = pos_emb_b;
            if (!pe_b) {
                RUN(t4r_dropout(stream, pos_emb, w.pe_b, nullptr, (long)B * 2 * L * D, 2L * L * D, drop_p, seed,
                                ctr_hi(offset, 255, SITE_POS)));
                pe_b = w.pe_b;
            }
            RUN(t4r_xlnet_kr_proj(stream, pe_b, w.planes, w.kr, (long)B * 2 * L, D));
        } else {
            RUN(t4r_xlnet_kr_proj(stream, pos_emb, w.planes, w.kr, 2L * L, D));
        }
        if (block) {
            // q | k | v projection, attention core, o-projection + dropout + residual + LayerNorm: ONE launch
            t4r_xlnet_attn_block_input_dropout(fuse_in, ctr_hi(offset, 255, SITE_INPUT), w.hin);
            const int rc_ab = t4r_xlnet_attn_block_fwd(stream, h, w.planes, params[P_O], w.kr, drop ? 2L * L * D : 0L, params[P_RWB],
                                                       params[P_RRB], params[P_LN1W], params[P_LN1B], w.qkv, w.av, w.lse, w.ao, w.mean1,
                                                       w.rstd1, w.h1, B, L, D, n_head, ln_eps, drop_p, seed, C(SITE_PROB),
                                                       C(SITE_ATTN_OUT), key_len);
            t4r_xlnet_attn_block_input_dropout(0, 0, nullptr);
            if (rc_ab) return rc_ab;
        } else {
        RUN(t4r_xlnet_attn_fwd(stream, w.qkv, w.qkv + TD, w.qkv + 2 * TD, w.kr, params[P_RWB], params[P_RRB], w.av, w.lse,
                               B, L, n_head, dh, drop, drop_p, seed, C(SITE_PROB), key_len));
        RUN(t4r_xlnet_oproj_ln(stream, w.av, h, w.planes, params[P_LN1W], params[P_LN1B], w.ao, w.mean1, w.rstd1, w.h1, T, D,
                               ln_eps, drop_p, seed, C(SITE_ATTN_OUT)));
        }
        const long ns = t4r_xlnet_ff_amax_slots(T);
        t4r_xlnet_ff_amax_buffers(w.amax, w.amax + ns, nullptr, nullptr);
        t4r_xlnet_ff_final_dropout(fuse_final && drop, ctr_hi(offset, 255, SITE_FINAL));
        const int rc = t4r_xlnet_ff_fwd(stream, w.h1, w.planes, params[P_B1], params[P_B2], params[P_LN2W], params[P_LN2B], w.ffpre,
                                        w.ffact, w.ffout, w.mean2, w.rstd2, h_out, T, D, ln_eps, drop_p, seed, C(SITE_FF_ACT),
                                        C(SITE_FF_OUT));
        t4r_xlnet_ff_final_dropout(0, 0);
        t4r_xlnet_ff_amax_buffers(nullptr, nullptr, nullptr, nullptr);
        return rc;
    }
    if (k_w == q_w + DD && v_w == q_w + 2 * DD) {
        RUN(t4r_gemm_launch(st, 0, 0, T, D, D, 1.f, h, D, q_w, D, w.qkv, D, nullptr, EPI_NONE, nullptr, 0,
                            1, 0, 3, 0, DD, TD, nullptr));
    } else {
        const float* ws3[3] = {q_w, k_w, v_w};
        for (int z = 0; z < 3; ++z)
            RUN(t4r_gemm_launch(st, 0, 0, T, D, D, 1.f, h, D, ws3[z], D, w.qkv + z * TD, D, nullptr,
                                EPI_NONE, nullptr, 0, 1, 0, 1, 0, 0, 0, nullptr));
    }
    if (drop) {
        // dropout(pos_emb expanded over the batch) @ r : per-session positional keys (HF :1142-1143
        // drops the batch-expanded pos_emb, so every session gets its own mask).
        // HF applies that dropout ONCE per forward, before the layer loop (:1143), and hands the same
        // dropped tensor to every layer: the mask is keyed by (offset, layer 255, SITE_POS), not by the layer.
        // pos_emb_b: that tensor [B, 2L, D], made once by the caller (t4r_dropout with this key) and shared by all
        // layers; NULL: this layer makes its own copy in its workspace (same values, one launch per layer)
        const float* pe_b = pos_emb_b;
        if (!pe_b) {
            RUN(t4r_dropout(stream, pos_emb, w.pe_b, nullptr, (long)B * 2 * L * D, 2L * L * D, drop_p, seed,
                            ctr_hi(offset, 255, SITE_POS)));
            pe_b = w.pe_b;
        }
        RUN(t4r_gemm_launch(st, 0, 0, B * 2 * L, D, D, 1.f, pe_b, D, params[P_R], D, w.kr, D, nullptr,
                            EPI_NONE, nullptr, 0, 1, 0, 1, 0, 0, 0, nullptr));
    } else {
        RUN(t4r_gemm_launch(st, 0, 0, 2 * L, D, D, 1.f, pos_emb, D, params[P_R], D, w.kr, D, nullptr,
                            EPI_NONE, nullptr, 0, 1, 0, 1, 0, 0, 0, nullptr));
    }
    RUN(t4r_xlnet_attn_fwd(stream, w.qkv, w.qkv + TD, w.qkv + 2 * TD, w.kr, params[P_RWB],
                           params[P_RRB], w.av, w.lse, B, L, n_head, dh, drop, drop_p, seed, C(SITE_PROB), key_len));
    // attn_out[t, h] = sum_{nd} av[t, nd] * o[h, nd]
    RUN(t4r_gemm_launch(st, 0, 1, T, D, D, 1.f, w.av, D, params[P_O], D, w.ao, D, nullptr, EPI_NONE,
                        nullptr, 0, 1, 0, 1, 0, 0, 0, nullptr));
    RUN(t4r_add_layernorm_fwd(stream, w.ao, h, params[P_LN1W], params[P_LN1B], w.h1, w.mean1, w.rstd1,
                              T, D, ln_eps, drop_p, seed, C(SITE_ATTN_OUT)));
    const DropCfg dff = make_drop(drop_p, seed, C(SITE_FF_ACT));
    RUN(t4r_gemm_launch(st, 0, 1, T, 4 * D, D, 1.f, w.h1, D, params[P_W1], D, w.ffact, 4 * D,
                        params[P_B1], EPI_BIAS_GELU, w.ffpre, 4 * D, 1, 0, 1, 0, 0, 0, &dff));
    RUN(t4r_gemm_launch(st, 0, 1, T, D, 4 * D, 1.f, w.ffact, 4 * D, params[P_W2], 4 * D, w.ffout, D,
                        params[P_B2], EPI_BIAS, nullptr, 0, 1, 0, 1, 0, 0, 0, nullptr));
    RUN(t4r_add_layernorm_fwd(stream, w.ffout, w.h1, params[P_LN2W], params[P_LN2B], h_out, w.mean2,
                              w.rstd2, T, D, ln_eps, drop_p, seed, C(SITE_FF_OUT)));
    return 0;
}

// Weight-gradient contractions feed nothing but the optimizer, so the layer backward issues them on a
// second HIP stream, under the kernels of the critical chain (LayerNorm / activation backward are
// HBM-bound, the attention core is latency-bound: the MFMA pipes are mostly idle there).
// Ordering is by events: the side stream waits for the producer of a wgrad's operands, the main
// stream waits (a) before a buffer a queued wgrad still reads is overwritten and (b) at the end of
// the call -- after the call returns, everything on `stream` is ordered after all of its work.
// T4R_LAYER_SIDE_STREAM=0 keeps everything on one stream.
struct SideStream {
    hipStream_t s = nullptr, s2 = nullptr;   // s2: the attention half's weight gradients (fused path), see t4r_xlnet_layer_bwd
    hipEvent_t done_s2 = nullptr, join1 = nullptr, join2 = nullptr;   // join*: recorded by t4r_xlnet_layer_bwd_join only (under its mutex)
    hipEvent_t fork[8] = {}, red[8] = {}, done_ff2 = nullptr, done_o = nullptr, done_all = nullptr;
    int state = 0;    // 0 untried, 1 ready, -1 disabled / failed
    int device = -1;
};
// One set per (host thread, device), heap-allocated and never freed: the registry below outlives the threads that
// created the entries (a worker thread that once ran a layer backward may exit; its streams stay valid and idle).
constexpr int kMaxDev = 16;
static thread_local SideStream* g_side[kMaxDev] = {};
// every thread's side streams (the autograd engine runs a layer backward on its device thread and its end-of-backward
// callback on whichever thread finishes the graph task: t4r_xlnet_layer_bwd_join waits for ALL of the CURRENT device)
#include <mutex>
#include <vector>
static std::mutex g_side_mu;
static std::vector<SideStream*> g_side_all;

static SideStream* side_stream() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
    if (!g_side[dev]) { g_side[dev] = new SideStream(); g_side[dev]->device = dev; }
    SideStream& ss = *g_side[dev];
    if (ss.state == 0) {
        const char* e = t4r_exp_getenv("T4R_LAYER_SIDE_STREAM");
        ss.state = -1;
        if (!(e && atoi(e) == 0)) {
            // lowest priority: the critical chain on the caller's stream gets the CUs first
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            const char* pe = t4r_exp_getenv("T4R_LAYER_SIDE_PRIO");
            const int prio = pe ? atoi(pe) : lo;
            // the events order streams of ONE device: no system-scope fence (cache write-back to host visibility) at each record
            const char* fe = t4r_exp_getenv("T4R_LAYER_EVENT_SYSFENCE");
            const unsigned evf = hipEventDisableTiming | ((fe && atoi(fe)) ? 0u : hipEventDisableSystemFence);
            bool ok = hipStreamCreateWithPriority(&ss.s, hipStreamNonBlocking, prio) == hipSuccess;
            ok = ok && hipStreamCreateWithPriority(&ss.s2, hipStreamNonBlocking, prio) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&ss.done_s2, evf) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&ss.join1, evf) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&ss.join2, evf) == hipSuccess;
            for (int i = 0; ok && i < 8; ++i) ok = hipEventCreateWithFlags(&ss.fork[i], evf) == hipSuccess;
            for (int i = 0; ok && i < 8; ++i) ok = hipEventCreateWithFlags(&ss.red[i], evf) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&ss.done_ff2, evf) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&ss.done_o, evf) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&ss.done_all, evf) == hipSuccess;
            if (ok) {
                ss.state = 1;
                std::lock_guard<std::mutex> lk(g_side_mu);
                g_side_all.push_back(&ss);
            }
        }
    }
    return ss.state == 1 ? &ss : nullptr;
}

// Deferred join (opt-in, t4r_xlnet_layer_bwd_defer(1)): the call returns WITHOUT making the caller's stream wait for its
// weight-gradient streams.  After the attention core only ~35 us of critical-chain work are left (d h) but ~90 us of
// weight-gradient work (r, q|k|v and their reduction can only start then): joined inside the call, the caller's stream
// idled 60-95 us at every layer boundary (rocprofv3 timeline, profiles/r03_g_*).  Deferred, the next layer's backward
// starts at once and that tail runs under it.  The caller then owes two things: every buffer the call was given
// (h, ws, bws, dh_out, pos_emb_b, the parameter and gradient tensors) stays alive and unwritten, and
// t4r_xlnet_layer_bwd_join(stream) is called before anything reads the parameter gradients or reuses those buffers
// (transformers4rec_amd/transformer.py does both from the autograd engine's end-of-backward callback).
static thread_local int g_defer_join = 0;
extern "C" void t4r_xlnet_layer_bwd_defer(int on) { g_defer_join = on ? 1 : 0; }

// grads[] are ACCUMULATED into (zero them / let the optimizer zero them between steps).
// dh_in [T,D] is overwritten with d loss / d h.
extern "C" int t4r_xlnet_layer_bwd(void* stream, const float* h, const float* pos_emb,
                                   const float* const* params, float* const* grads, const float* ws,
                                   float* bws, const float* dh_out, float* dh_in, int B, int L, int D,
                                   int n_head, float ln_eps, float drop_p, unsigned long long seed,
                                   unsigned long long offset, int layer_idx, const int* key_len,
                                   const float* pos_emb_b) {
    if (B == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int T = B * L, dh = D / n_head;
    const long TD = (long)T * D, DD = (long)D * D;
    const int drop = drop_p > 0.f;
    const long nkr = (drop ? (long)B : 1L) * 2L * L * D;
    LayerWs w = carve(const_cast<float*>(ws), B, L, D, n_head, drop);
    auto C = [&](int site) { return ctr_hi(offset, layer_idx, site); };
    const bool fuse_final = (layer_idx & T4R_LAYER_FUSE_FINAL) != 0;
    T4R_CHECK_ARG(!fuse_final || use_fused(D), "xlnet_layer: the fused output dropout needs the fused layer kernels (d_model 32 / 64 / 128)");
    const bool fuse_in = (layer_idx & T4R_LAYER_FUSE_INPUT) != 0 && drop;
    T4R_CHECK_ARG(!fuse_in || (use_fused(D) && use_attn_block(L, D, n_head)),
                  "xlnet_layer: the fused input dropout needs the one-kernel attention forward (L <= 32, d_model 32 / 64 / 128)");
    if (fuse_in) h = w.hin;          // the forward left the DROPPED input rows there: the residual and the weight-gradient operand
    long o = 0;
    auto take = [&](long nfl) { float* p = bws + o; o += align4(nfl); return p; };
    float* dqkv = take(3 * TD);
    float* dav = take(TD);
    float* dx = take(TD);
    float* dff = take(4 * TD);
    float* dkr = take(nkr);
    float* attn_ws = take(t4r_xlnet_attn_bwd_ws_floats(B, L, D, n_head));
    // one partial buffer per column-reduction site: their second stages run on the side stream, possibly
    // after the next site's first stage
    float* red_act = take(t4r_colreduce_ws_floats(T, 4 * D));
    float* red_ln2 = take(t4r_colreduce_ws_floats(T, 2 * D));
    float* red_ln1 = take(t4r_colreduce_ws_floats(T, 2 * D));
    float* red_b2 = take(t4r_colreduce_ws_floats(T, D));
    float* dxa = drop ? take(TD) : nullptr;     // gradient of a dropped LayerNorm operand
    float* dfo = take(TD);                      // fused feed-forward backward: d ffout rows (own buffer: the FF2 weight
                                                // gradient may still read them while the attention half runs)
    float* ff_part = take(t4r_xlnet_ff_bwd_part_floats(T, D));
    float* dao_buf = take(TD);                  // fused LayerNorm-1 backward: d attn_out rows (own buffer, as dfo)
    float* ln1_part = take(t4r_xlnet_ln1_bwd_part_floats(T, D));
#ifdef T4R_EXPERIMENTAL
    float* block_part = take(attn_block_bwd_part(B, L, D, n_head));
#endif
    const bool fused = use_fused(D);
    // the split-K weight gradients of this call leave their partial tiles here; one launch adds them at the end
    struct SinkGuard {
        SinkGuard(float* ws, long cap) { t4r_splitk_sink_begin(ws, cap); }
        ~SinkGuard() { t4r_splitk_sink_end(); }
    } sink_guard(take(splitk_sink_floats(T, D)), splitk_sink_floats(T, D));

    SideStream* ss = side_stream();
    struct RedirectGuard {      // second stages of the column reductions -> side stream, for this call only
        bool on;
        RedirectGuard(SideStream* s) : on(false) {
            static const int enabled = [] { const char* e = t4r_exp_getenv("T4R_LAYER_SIDE_REDUCE"); return e ? atoi(e) : 1; }();
            if (s && enabled) { t4r_reduce_redirect(s->s, s->red, 8); on = true; }
        }
        ~RedirectGuard() { if (on) t4r_reduce_redirect(nullptr, nullptr, 0); }
    } redirect(ss);
    int n_fork = 0;
    // wg(): the stream a weight-gradient GEMM goes to; it first waits for everything issued on `st` so far
    auto wg = [&]() -> hipStream_t {
        if (!ss) return st;
        (void)hipEventRecord(ss->fork[n_fork], st);
        (void)hipStreamWaitEvent(ss->s, ss->fork[n_fork], 0);
        ++n_fork;
        return ss->s;
    };
    // wg_again(): the first side stream once more, for a product whose operands were complete at the previous wg() already
    // (every event recorded on the caller's stream costs it ~6 us before its next kernel starts: one per group of products)
    auto wg_again = [&]() -> hipStream_t { return ss ? ss->s : st; };
    // wg2(): the second weight-gradient stream.  One side stream is a serial queue: the o / r / q / k / v products (ready
    // when LayerNorm-1 backward and the attention core are done) sat behind the two feed-forward products and their tail
    // -- five 10-25 us kernels with ~6 us between them -- ran AFTER the critical chain had finished: ~95 us of idle
    // caller's stream at every layer boundary (rocprofv3 timeline, 4 x per step).  On their own stream they start when
    // their operands exist.
    static const int two_side = [] { const char* e = t4r_exp_getenv("T4R_LAYER_SIDE_STREAMS"); return e ? atoi(e) : 2; }();
    bool used_s2 = false;
    auto wg2 = [&]() -> hipStream_t {
        if (!ss) return st;
        if (two_side < 2) return wg();
        (void)hipEventRecord(ss->fork[n_fork], st);
        (void)hipStreamWaitEvent(ss->s2, ss->fork[n_fork], 0);
        ++n_fork;
        used_s2 = true;
        return ss->s2;
    };

    if (fused) {
        // one launch: LayerNorm backward -> d ffout -> FF2 dX -> GELU' / dropout -> FF1 dX + residual (+ the partial
        // sums of d gamma, d beta, d b2, d b1, reduced by two small launches on the weight-gradient stream)
        const long ns = t4r_xlnet_ff_amax_slots(T);
        // workgroups (= maxima slots) of the forward launches and of this backward launch: the backward's tile follows the CU
        // budget (t4r_xlnet_set_cu_budget: a collective may hold CUs now), the forward's never does
        const int na = t4r_xlnet_ff_amax_count(T), nb = t4r_xlnet_ff_amax_count_bwd(T);
        float* am = w.amax;          // max |h1|, max |act| (forward), max |d pre|, max |d ffout| (now): one float per workgroup
        t4r_xlnet_ff_amax_buffers(nullptr, nullptr, am + 2 * ns, am + 3 * ns);
        {
            t4r_xlnet_ff_final_dropout(fuse_final && drop, ctr_hi(offset, 255, SITE_FINAL));
            const int rc = t4r_xlnet_ff_bwd(stream, dh_out, w.ffout, w.h1, w.mean2, w.rstd2, params[P_LN2W], w.ffpre, w.planes,
                                            dx, dfo, dff, grads[P_LN2W], grads[P_LN2B], grads[P_B2], grads[P_B1], ff_part,
                                            T, D, drop_p, seed, C(SITE_FF_ACT), C(SITE_FF_OUT));
            t4r_xlnet_ff_final_dropout(0, 0);
            t4r_xlnet_ff_amax_buffers(nullptr, nullptr, nullptr, nullptr);
            if (rc) return rc;
        }
        // the two weight gradients in the two-way fp16 form, positioned by the operand maxima the fused kernels left
        const bool hs = t4r_xlnet_body_fp16x2() && drop_p >= 0.f;
        if (hs) t4r_gemm_operand_amax2(am + 3 * ns, nb, am + ns, na);
        RUN(t4r_gemm_launch(wg(), 1, 0, D, 4 * D, T, 1.f, dfo, D, w.ffact, 4 * D, grads[P_W2], 4 * D, nullptr,
                            EPI_NONE, nullptr, 0, -1, 1, 1, 0, 0, 0, nullptr));
        {
            hipStream_t s1 = wg_again();
            if (hs) t4r_gemm_operand_amax2(am + 2 * ns, nb, am, na);
            RUN(t4r_gemm_launch(s1, 1, 0, 4 * D, D, T, 1.f, dff, 4 * D, w.h1, D, grads[P_W1], D, nullptr,
                                EPI_NONE, nullptr, 0, -1, 1, 1, 0, 0, 0, nullptr));
            // first reduction launch of the call: W2, W1 and the bias / LayerNorm-2 sums of the feed-forward half
            RUN(t4r_splitk_sink_flush(s1));
        }
    } else {
    // LN2: y = LN(drop(ffout) + h1): dx = d h1 (residual part), d ffout = dxa (or dx when p = 0)
    RUN(t4r_add_layernorm_bwd(stream, w.ffout, w.h1, params[P_LN2W], w.mean2, w.rstd2, dh_out, dx, dxa,
                              grads[P_LN2W], grads[P_LN2B], red_ln2, T, D, 0, drop_p, seed, C(SITE_FF_OUT)));
    const float* dffout = drop ? dxa : dx;
    // FF2: ffout = ffact @ w2^T + b2
    RUN(t4r_gemm_launch(st, 0, 0, T, 4 * D, D, 1.f, dffout, D, params[P_W2], 4 * D, dff, 4 * D, nullptr,
                        EPI_NONE, nullptr, 0, 1, 0, 1, 0, 0, 0, nullptr));
    RUN(t4r_colsum(stream, dffout, grads[P_B2], red_b2, T, D, D));
    RUN(t4r_gemm_launch(wg(), 1, 0, D, 4 * D, T, 1.f, dffout, D, w.ffact, 4 * D, grads[P_W2], 4 * D, nullptr,
                        EPI_NONE, nullptr, 0, -1, 1, 1, 0, 0, 0, nullptr));
    if (ss) (void)hipEventRecord(ss->done_ff2, ss->s);
    // ffact = drop(gelu(ffpre)) ; GELU' + bias1
    RUN(t4r_act_bwd_bias(stream, dff, w.ffpre, dff, grads[P_B1], red_act, T, 4 * D, 0, drop_p, seed,
                         C(SITE_FF_ACT)));
    // FF1: ffpre = h1 @ w1^T + b1 ;  d h1 = dx (residual) + dff @ w1
    if (ss && !drop) (void)hipStreamWaitEvent(st, ss->done_ff2, 0);   // p = 0: the FF2 wgrad reads dx, written next
    RUN(t4r_gemm_launch(st, 0, 0, T, D, 4 * D, 1.f, dff, 4 * D, params[P_W1], D, dx, D, nullptr,
                        EPI_NONE, nullptr, 0, 1, 1, 1, 0, 0, 0, nullptr));
    RUN(t4r_gemm_launch(wg(), 1, 0, 4 * D, D, T, 1.f, dff, 4 * D, w.h1, D, grads[P_W1], D, nullptr,
                        EPI_NONE, nullptr, 0, -1, 1, 1, 0, 0, 0, nullptr));
    }
    if (fused) {
#ifdef T4R_EXPERIMENTAL
        const bool block = use_attn_block_bwd(L, D, n_head);
        if (block) {
            // LayerNorm-1 backward, d attn_vec, the attention core backward and d h: ONE launch (tools/experimental)
            RUN(t4r_xlnet_attn_block_bwd(stream, dx, w.ao, h, w.mean1, w.rstd1, params[P_LN1W], w.planes, params[P_Q], params[P_K],
                                         params[P_V], w.qkv, w.kr, drop ? 2L * L * D : 0L, params[P_RWB], params[P_RRB], w.lse,
                                         dh_in, dao_buf, dqkv, dkr, grads[P_RWB], grads[P_RRB], grads[P_LN1W], grads[P_LN1B],
                                         block_part, B, L, D, n_head, drop_p, seed, C(SITE_PROB), C(SITE_ATTN_OUT), key_len));
        } else
#else
        constexpr bool block = false;
#endif
        {
        // LayerNorm-1 backward + d attn_vec in one launch; the attention core; d h from d q, d k, d v in one launch
        RUN(t4r_xlnet_ln1_bwd(stream, dx, w.ao, h, w.mean1, w.rstd1, params[P_LN1W], w.planes, dh_in, dao_buf, dav,
                              grads[P_LN1W], grads[P_LN1B], ln1_part, T, D, drop_p, seed, C(SITE_ATTN_OUT)));
        RUN(t4r_xlnet_attn_bwd(stream, w.qkv, w.qkv + TD, w.qkv + 2 * TD, w.kr, params[P_RWB],
                               params[P_RRB], w.av, w.lse, dav, dqkv, dqkv + TD, dqkv + 2 * TD, dkr,
                               grads[P_RWB], grads[P_RRB], attn_ws, B, L, n_head, dh, drop, drop_p, seed,
                               C(SITE_PROB), key_len));
        }
        // ONE event for the o, r and q|k|v products (their operands are all complete here; with the join deferred nothing
        // is gained by starting the o product before the attention core)
        hipStream_t sw = wg2();
        RUN(t4r_gemm_launch(sw, 1, 0, D, D, T, 1.f, dao_buf, D, w.av, D, grads[P_O], D, nullptr, EPI_NONE,
                            nullptr, 0, -1, 1, 1, 0, 0, 0, nullptr));
        if (drop) {
            RUN(t4r_gemm_launch(sw, 1, 0, D, D, B * 2 * L, 1.f, pos_emb_b ? pos_emb_b : w.pe_b, D, dkr, D, grads[P_R], D,
                                nullptr, EPI_NONE, nullptr, 0, -1, 1, 1, 0, 0, 0, nullptr));
        } else {
            // shared k_r: d k_r is finished by a reduction that was redirected to the FIRST side stream -- its consumer follows it there
            RUN(t4r_gemm_launch(wg(), 1, 0, D, D, 2 * L, 1.f, pos_emb, D, dkr, D, grads[P_R], D, nullptr,
                                EPI_NONE, nullptr, 0, 1, 1, 1, 0, 0, 0, nullptr));
        }
        {
            float* gz[3] = {grads[P_Q], grads[P_K], grads[P_V]};
            if (gz[1] == gz[0] + DD && gz[2] == gz[0] + 2 * DD) {
                RUN(t4r_gemm_launch(sw, 1, 0, D, D, T, 1.f, h, D, dqkv, D, gz[0], D, nullptr, EPI_NONE, nullptr,
                                    0, -1, 1, 3, 0, TD, DD, nullptr));
            } else {
                for (int z = 0; z < 3; ++z)
                    RUN(t4r_gemm_launch(sw, 1, 0, D, D, T, 1.f, h, D, dqkv + z * TD, D, gz[z], D, nullptr,
                                        EPI_NONE, nullptr, 0, -1, 1, 1, 0, 0, 0, nullptr));
            }
        }
        // second reduction launch: o, r, q, k, v and the LayerNorm-1 / attention-bias sums (same stream as their products)
        RUN(t4r_splitk_sink_flush(sw));
        if (!block) {
            t4r_xlnet_dh_input_dropout(fuse_in, drop_p, seed, ctr_hi(offset, 255, SITE_INPUT));
            const int rc_dh = t4r_xlnet_dh(stream, dqkv, w.planes, dh_in, T, D);
            t4r_xlnet_dh_input_dropout(0, 0.f, 0, 0);
            if (rc_dh) return rc_dh;
        }
        if (ss && !g_defer_join) {   // join: the caller's stream continues after every weight gradient of this layer
            (void)hipEventRecord(ss->done_all, ss->s);
            (void)hipStreamWaitEvent(st, ss->done_all, 0);
            if (used_s2) {
                (void)hipEventRecord(ss->done_s2, ss->s2);
                (void)hipStreamWaitEvent(st, ss->done_s2, 0);
            }
        }
        return 0;
    }
    // LN1: h1 = LN(drop(ao) + h): dh_in = d h (residual part), d ao = dxa (or dh_in when p = 0)
    if (ss && drop) (void)hipStreamWaitEvent(st, ss->done_ff2, 0);    // the FF2 wgrad reads dxa, overwritten here
    RUN(t4r_add_layernorm_bwd(stream, w.ao, h, params[P_LN1W], w.mean1, w.rstd1, dx, dh_in, dxa,
                              grads[P_LN1W], grads[P_LN1B], red_ln1, T, D, 0, drop_p, seed, C(SITE_ATTN_OUT)));
    const float* dao = drop ? dxa : dh_in;
    // O projection: ao = av @ o^T
    RUN(t4r_gemm_launch(st, 0, 0, T, D, D, 1.f, dao, D, params[P_O], D, dav, D, nullptr, EPI_NONE,
                        nullptr, 0, 1, 0, 1, 0, 0, 0, nullptr));
    RUN(t4r_gemm_launch(wg(), 1, 0, D, D, T, 1.f, dao, D, w.av, D, grads[P_O], D, nullptr, EPI_NONE,
                        nullptr, 0, -1, 1, 1, 0, 0, 0, nullptr));
    if (ss) (void)hipEventRecord(ss->done_o, ss->s);
    // attention core
    RUN(t4r_xlnet_attn_bwd(stream, w.qkv, w.qkv + TD, w.qkv + 2 * TD, w.kr, params[P_RWB],
                           params[P_RRB], w.av, w.lse, dav, dqkv, dqkv + TD, dqkv + 2 * TD, dkr,
                           grads[P_RWB], grads[P_RRB], attn_ws, B, L, n_head, dh, drop, drop_p, seed,
                           C(SITE_PROB), key_len));
    // k_r = pos_emb(_b) @ r  ->  d r += pos_emb(_b)^T @ d k_r
    if (drop) {
        RUN(t4r_gemm_launch(wg(), 1, 0, D, D, B * 2 * L, 1.f, pos_emb_b ? pos_emb_b : w.pe_b, D, dkr, D, grads[P_R], D, nullptr,
                            EPI_NONE, nullptr, 0, -1, 1, 1, 0, 0, 0, nullptr));
    } else {
        RUN(t4r_gemm_launch(wg(), 1, 0, D, D, 2 * L, 1.f, pos_emb, D, dkr, D, grads[P_R], D, nullptr,
                            EPI_NONE, nullptr, 0, 1, 1, 1, 0, 0, 0, nullptr));
    }
    // q,k,v = h @ w  ->  d h += d{q,k,v} @ w^T ; d w += h^T @ d{q,k,v}
    const float* wz[3] = {params[P_Q], params[P_K], params[P_V]};
    float* gz[3] = {grads[P_Q], grads[P_K], grads[P_V]};
    {
        hipStream_t sw = wg();
        if (gz[1] == gz[0] + DD && gz[2] == gz[0] + 2 * DD) {
            RUN(t4r_gemm_launch(sw, 1, 0, D, D, T, 1.f, h, D, dqkv, D, gz[0], D, nullptr, EPI_NONE, nullptr,
                                0, -1, 1, 3, 0, TD, DD, nullptr));
        } else {
            for (int z = 0; z < 3; ++z)
                RUN(t4r_gemm_launch(sw, 1, 0, D, D, T, 1.f, h, D, dqkv + z * TD, D, gz[z], D, nullptr,
                                    EPI_NONE, nullptr, 0, -1, 1, 1, 0, 0, 0, nullptr));
        }
    }
    if (ss && !drop) (void)hipStreamWaitEvent(st, ss->done_o, 0);     // p = 0: the O wgrad reads dh_in, accumulated into next
    for (int z = 0; z < 3; ++z)
        RUN(t4r_gemm_launch(st, 0, 1, T, D, D, 1.f, dqkv + z * TD, D, wz[z], D, dh_in, D, nullptr,
                            EPI_NONE, nullptr, 0, 1, 1, 1, 0, 0, 0, nullptr));
    RUN(t4r_splitk_sink_flush(wg()));      // after every producer of a partial on the caller's stream
    if (ss) {   // join: the caller's stream continues after every weight gradient of this layer
        (void)hipEventRecord(ss->done_all, ss->s);
        (void)hipStreamWaitEvent(st, ss->done_all, 0);
    }
    return 0;
}

// the caller's stream waits for everything queued so far on the weight-gradient streams of the current device, whichever
// host thread queued it (no-op without them)
extern "C" int t4r_xlnet_layer_bwd_join(void* stream) {
    hipStream_t st = (hipStream_t)stream;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { t4r_set_error("xlnet_layer_bwd_join: no current device"); return -1; }
    std::lock_guard<std::mutex> lk(g_side_mu);
    for (SideStream* ss : g_side_all) {
        if (ss->device != dev) continue;
        (void)hipEventRecord(ss->join1, ss->s);
        (void)hipStreamWaitEvent(st, ss->join1, 0);
        (void)hipEventRecord(ss->join2, ss->s2);
        (void)hipStreamWaitEvent(st, ss->join2, 0);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
#include <string.h>
static thread_local char g_err[512] = "";
extern "C" void t4r_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* t4r_last_error(void) { return g_err; }
extern "C" int t4r_abi_version(void) { return 1; }
