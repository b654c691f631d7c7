// Input block of the session-sequence path: multi-feature embedding gather + aggregation,
// SoftEmbedding (+ per-feature LayerNorm), optional masking epilogue, and the ragged->padded
// conversion in front of it.  HBM-bound integer/copy work: one fused launch for all tables,
// 16-byte row accesses, several token rows in flight per wave.
//
// Reference behaviour restated (paths relative to the reference repo):
//   SequenceEmbeddingFeatures / EmbeddingFeatures.forward   transformers4rec/torch/features/embedding.py:226-249
//       (row 0 of a table is a normal, randomly initialised row -- features/sequence.py:75-81;
//        padding_idx=0 only blocks the *gradient* of row 0)
//   SoftEmbedding.forward                                   features/embedding.py:551-556
//   SoftEmbeddingFeatures post LayerNorm (eps 1e-5)         features/embedding.py:306-309
//   ConcatFeatures (sorted feature names -> column offsets chosen by the host)
//                                                           torch/tabular/aggregation.py:35-47
//   ElementwiseSum / ElementwiseSumItemMulti                torch/tabular/aggregation.py:140-193
//   MaskSequence.apply_mask_to_inputs (MLM / CLM variants)  torch/masking.py:302-337,473-498
//   _pad_ragged_tensor / pad_inputs                         torch/utils/padding.py:48-68,126-164
#include "t4r_common.h"

#define T4R_MAX_FEATS 16
#define T4R_SOFT_MAXK 32
#define T4R_SOFT_MAXD 32

enum { AGG_CONCAT = 0, AGG_SUM = 1, AGG_SUM_ITEM_MULTI = 2 };
enum { MASK_NONE = 0, MASK_MLM = 1, MASK_CLM = 2, MASK_CLM_INFER = 3 };

struct SeqFeatParams {
    int n_feat;
    int kind[T4R_MAX_FEATS];          // 0 table lookup, 1 dense rows (precomputed), 2 per-session table lookup,
                                      // 3 per-session dense rows
    const void* input[T4R_MAX_FEATS]; // kind 0: int64 ids [B*L_in] ; 1: float [B*L_in, dim] ; 2: int64 ids [B] ;
                                      // 3: float [B, dim]
    const float* table[T4R_MAX_FEATS];
    int dim[T4R_MAX_FEATS];
    int col[T4R_MAX_FEATS];           // output column offset (concat) / 0 (sum)
    long rows[T4R_MAX_FEATS];         // table rows (bounds check)
    int agg, item_feat;               // item_feat: index of the item-id feature (item-multi)
    int B, L_in, L_out, W;            // W = output row width
    // masking epilogue (only legal when no projection follows, i.e. W == hidden)
    int mask_mode;
    const unsigned char* mask;        // [B*L_out]
    const float* masked_emb;          // [W]
    float* out;                       // [B*L_out, W]
    int* err;                         // set to 1 on an out-of-range id
};

// One token row per GROUP lanes (GROUP = power of two <= 64); each lane produces 4 consecutive
// output columns per step.  "Dense" features (soft embeddings computed by
// t4r_soft_embedding_fwd, continuous pass-through) are copied from their [tokens, dim] rows.
template <int GROUP>
__global__ __launch_bounds__(256) void seq_features_fwd_kernel(SeqFeatParams p) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int gl = threadIdx.x & (GROUP - 1);
    const long tok = (long)tid / GROUP;                 // output token index b*L_out + l
    const long ntok = (long)p.B * p.L_out;
    if (tok >= ntok) return;
    const int b = (int)(tok / p.L_out), l = (int)(tok % p.L_out);
    const int ls = min(l, p.L_in - 1);                  // MLM inference: position L duplicates L-1
    const long ts = (long)b * p.L_in + ls;

    int mode = p.mask_mode;
    bool m = false;
    if (mode != MASK_NONE) m = p.mask[tok] != 0;
    float* orow = p.out + tok * p.W;

    for (int c0 = gl * 4; c0 < p.W; c0 += GROUP * 4) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        // masking decides whether the embedding is needed at all
        bool use_emb = true, zero = false;
        if (mode == MASK_MLM) use_emb = !m;
        else if (mode == MASK_CLM) { use_emb = m; zero = (l == p.L_out - 1); }
        else if (mode == MASK_CLM_INFER) use_emb = m;
        if (use_emb && !zero) {
            float other[4] = {0.f, 0.f, 0.f, 0.f};
            float item[4] = {0.f, 0.f, 0.f, 0.f};
            for (int f = 0; f < p.n_feat; ++f) {
                int lc0;  // first column inside feature f handled by this lane, or skip
                if (p.agg == AGG_CONCAT) {
                    if (c0 + 4 <= p.col[f] || c0 >= p.col[f] + p.dim[f]) continue;
                    lc0 = c0 - p.col[f];
                } else {
                    lc0 = c0;
                }
                float fv[4] = {0.f, 0.f, 0.f, 0.f};
                const float* row;
                if (p.kind[f] == 0 || p.kind[f] == 2) {
                    // kind 2: a non-sequential (context) feature, one id per session, broadcast over
                    // the sequence (ConcatFeatures._expand_non_sequential_features, tabular/base.py:53-63)
                    long id = reinterpret_cast<const long*>(p.input[f])[p.kind[f] == 2 ? (long)b : ts];
                    if (id < 0 || id >= p.rows[f]) { if (p.err) *p.err = 1; id = 0; }
                    row = p.table[f] + id * p.dim[f];
                } else {
                    // kind 1: dense rows per token ; kind 3: dense rows per session (a context feature
                    // that went through a post transformation), broadcast over the sequence
                    row = reinterpret_cast<const float*>(p.input[f]) + (p.kind[f] == 3 ? (long)b : ts) * p.dim[f];
                }
                if (lc0 >= 0 && lc0 + 4 <= p.dim[f] && (p.dim[f] & 3) == 0) {
                    const float4 t = *reinterpret_cast<const float4*>(row + lc0);
                    fv[0] = t.x; fv[1] = t.y; fv[2] = t.z; fv[3] = t.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = lc0 + e;
                        if (c >= 0 && c < p.dim[f]) fv[e] = row[c];
                    }
                }
                if (p.agg == AGG_CONCAT) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = lc0 + e;
                        if (c >= 0 && c < p.dim[f]) v[e] = fv[e];
                    }
                } else if (p.agg == AGG_SUM) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += fv[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (f == p.item_feat) item[e] = fv[e]; else other[e] += fv[e];
                    }
                }
            }
            if (p.agg == AGG_SUM_ITEM_MULTI) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = item[e] * other[e];
            }
        } else if (!use_emb) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c0 + e < p.W) v[e] = p.masked_emb[c0 + e];
        }
        if (c0 + 4 <= p.W && (p.W & 3) == 0) {
            *reinterpret_cast<float4*>(orow + c0) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c0 + e < p.W) orow[c0 + e] = v[e];
        }
    }
}

// Concatenation fast path: every feature width and column offset a multiple of 4, so a lane owns fixed
// (feature, 16-byte chunk) pairs for all its tokens -- NCH chunks per lane (chunk j of lane gl = columns
// (gl + GROUP*j)*4 ..), i.e. rows up to GROUP*NCH*4 = 1024 floats wide (C3's 336-wide concatenation of
// item 128 + 3 x 64 + 2 x 8 soft-embedding rows takes GROUP 64, NCH 2).  U consecutive tokens per lane group
// with all id loads, then all row loads, in flight together (the generic kernel has one dependent
// id -> row chain per lane).
// NT: the table rows are read with non-temporal loads.  Chosen by the host when EVERY gathered table is far beyond the 256 MB
// Infinity Cache (rows touched once per launch should not evict ids / output lines); with small tables in the row (C3's three
// categoricals: cache resident, hit again and again) plain loads win.  Measured (tools/gather_sweep.py, 163 840 tokens, 10 M-row
// item table; box copy ceiling 5.1-5.2 TB/s): item-only 0.66 -> 0.70 of 8 TB/s with NT; C3 0.659 plain, 0.643 NT on every
// table, 0.48 with a per-feature run-time choice (two load forms in one wave: the loads serialise).  T4R_GATHER_NT overrides.
template <int GROUP, int U, int NCH, bool NT = false>
__global__ __launch_bounds__(256) void seq_features_fwd_fast_kernel(SeqFeatParams p) {
    const int gl = threadIdx.x & (GROUP - 1);
    const int grp = (int)(((long)blockIdx.x * 256 + threadIdx.x) / GROUP);
    const int ntok = p.B * p.L_out;
    const int tok0 = grp * U;
    if (gl * 4 >= p.W || tok0 >= ntok) return;
    int kind[NCH], dim[NCH], lc0[NCH];
    long rows[NCH];
    const void* input[NCH];
    const float* table[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c0 = (gl + GROUP * j) * 4;
        kind[j] = -1; dim[j] = 4; lc0[j] = 0; rows[j] = 0; input[j] = nullptr; table[j] = nullptr;
        for (int f = 0; f < p.n_feat; ++f) {
            if (c0 < p.W && c0 >= p.col[f] && c0 < p.col[f] + p.dim[f]) {
                kind[j] = p.kind[f]; dim[j] = p.dim[f]; lc0[j] = c0 - p.col[f]; rows[j] = p.rows[f];
                input[j] = p.input[f]; table[j] = p.table[f];
            }
        }
    }
    const int mode = p.mask_mode;
    bool use[U], zero[U];
    long id[U][NCH];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int tok = min(tok0 + u, ntok - 1);
        const int b = tok / p.L_out, l = tok - b * p.L_out;
        const int ts = b * p.L_in + min(l, p.L_in - 1);
        const bool m = mode != MASK_NONE && p.mask[tok] != 0;
        use[u] = true; zero[u] = false;
        if (mode == MASK_MLM) use[u] = !m;
        else if (mode == MASK_CLM) { use[u] = m; zero[u] = (l == p.L_out - 1); }
        else if (mode == MASK_CLM_INFER) use[u] = m;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const long r = (kind[j] >= 2) ? (long)b : (long)ts;
            id[u][j] = r;
            if (kind[j] == 0 || kind[j] == 2) id[u][j] = reinterpret_cast<const long*>(input[j])[r];
        }
    }
    const float* src[U][NCH];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (kind[j] == 0 || kind[j] == 2) {
                if (id[u][j] < 0 || id[u][j] >= rows[j]) { if (p.err) *p.err = 1; id[u][j] = 0; }
                src[u][j] = table[j] + id[u][j] * dim[j] + lc0[j];
            } else {
                src[u][j] = reinterpret_cast<const float*>(input[j]) + id[u][j] * dim[j] + lc0[j];
            }
        }
    float4 v[U][NCH];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            v[u][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kind[j] < 0) continue;
            if (use[u] && !zero[u]) {
                if constexpr (NT) {
                    typedef float ntl4 __attribute__((ext_vector_type(4)));
                    const ntl4 t = __builtin_nontemporal_load(reinterpret_cast<const ntl4*>(src[u][j]));
                    v[u][j] = make_float4(t[0], t[1], t[2], t[3]);
                } else {
                    v[u][j] = *reinterpret_cast<const float4*>(src[u][j]);
                }
            }
            else if (!use[u]) v[u][j] = *reinterpret_cast<const float4*>(p.masked_emb + (gl + GROUP * j) * 4);
        }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < NCH; ++j)
            if (tok0 + u < ntok && kind[j] >= 0) {
                // streaming store: the output is consumed by the next kernel from HBM/MALL anyway, and a write-allocating
                // store evicts table rows that other lookups of this launch would still hit in L2
                typedef float nt4 __attribute__((ext_vector_type(4)));
                const nt4 o = {v[u][j].x, v[u][j].y, v[u][j].z, v[u][j].w};
                __builtin_nontemporal_store(o, reinterpret_cast<nt4*>(p.out + (long)(tok0 + u) * p.W + (gl + GROUP * j) * 4));
            }
}

static int pick_group(int W) {
    int units = (W + 3) / 4, g = 1;
    while (g < units && g < 64) g <<= 1;
    return g < 8 ? 8 : g;
}

extern "C" int t4r_seq_features_fwd(
    void* stream, int n_feat, const int* kind, const void* const* input, const float* const* table,
    const int* dim, const int* col, const long* rows, int agg, int item_feat, int B, int L_in,
    int L_out, int W, int mask_mode, const unsigned char* mask, const float* masked_emb, float* out,
    int* err_flag) {
    T4R_CHECK_ARG(n_feat >= 1 && n_feat <= T4R_MAX_FEATS, "seq_features: 1..16 features");
    T4R_CHECK_ARG(B >= 0 && L_in >= 1 && L_out >= L_in && W >= 1, "seq_features: bad shape");
    T4R_CHECK_ARG(mask_mode == MASK_NONE || (mask && masked_emb), "seq_features: mask inputs missing");
    if (B == 0) return 0;
    SeqFeatParams p;
    p.n_feat = n_feat;
    for (int f = 0; f < n_feat; ++f) {
        p.kind[f] = kind[f]; p.input[f] = input[f]; p.table[f] = table[f]; p.dim[f] = dim[f];
        p.col[f] = col ? col[f] : 0; p.rows[f] = rows ? rows[f] : 0;
        T4R_CHECK_ARG(input[f] && (kind[f] == 1 || kind[f] == 3 || table[f]), "seq_features: null feature pointer");
        T4R_CHECK_ARG(kind[f] >= 0 && kind[f] <= 3, "seq_features: feature kind 0..3");
        if (agg != AGG_CONCAT) T4R_CHECK_ARG(dim[f] == W, "seq_features: element-wise needs equal dims");
    }
    p.agg = agg; p.item_feat = item_feat;
    p.B = B; p.L_in = L_in; p.L_out = L_out; p.W = W;
    p.mask_mode = mask_mode; p.mask = mask; p.masked_emb = masked_emb; p.out = out; p.err = err_flag;
    // Concatenation fast path, U = 2 tokens per lane group (T4R_GATHER_U: 0 = generic kernel, 4 = four
    // tokens).  HIP-graph replay, no host overhead (tools/gather_bench.py), item table 100 001 x 128:
    //   tokens      generic           U = 2             U = 4
    //   20 480      6.5 us (41 %)     5.0 us (53 %)     5.5 us (48 %)      of the 8 TB/s HBM peak
    //   163 840     30.8 us (69 %)    26.7 us (79 %)    27.7 us (76 %)
    //   1 310 720   252 us (67 %)     198 us (85 %)     191 us (89 %)
    // (10 M-row table: 66 % -> 76 % at 163 840 tokens, 61 % -> 67 % at 1.3 M tokens)
    const int g = pick_group(W);
    static int fast_u = -1;
    static bool fast_u_set = false;
    if (fast_u < 0) { const char* e = t4r_exp_getenv("T4R_GATHER_U"); fast_u_set = e != nullptr; fast_u = e ? atoi(e) : 2; }
    const int units = (W + 3) / 4;
    const int nch = (units + g - 1) / g;                // 16-byte chunks per lane
    bool fast_ok = fast_u > 0 && agg == AGG_CONCAT && (W & 3) == 0 && nch <= 4 && g >= 8 &&
                   (long)B * L_out * g < 0x7fffffffL;
    for (int f = 0; f < n_feat && fast_ok; ++f) fast_ok = (p.dim[f] & 3) == 0 && (p.col[f] & 3) == 0;
    if (fast_ok) {
        // tokens per lane group: 2 for rows up to 256 floats (table above); 4 for the two-chunk rows (C3's 336-wide
        // concatenation: 61 % -> 72 % of 8 TB/s at 163 840 tokens, 58 % -> 78 % at 1.3 M with a 10 M-row item table,
        // profiles/r02_e_c3_gather.txt); T4R_GATHER_U overrides
        const int U = nch == 2 ? ((fast_u_set && fast_u < 4) ? 2 : 4) : ((fast_u >= 4 && nch == 1) ? (fast_u >= 8 ? 8 : 4) : 2);
        // non-temporal row loads when a gathered table cannot stay on the chip anyway (> 256 MB: beyond the Infinity Cache)
        static int nt_env = -2;
        if (nt_env == -2) { const char* e = t4r_exp_getenv("T4R_GATHER_NT"); nt_env = e ? atoi(e) : -1; }
        bool nt = true, any = false;
        for (int f = 0; f < n_feat; ++f)
            if (p.kind[f] == 0 || p.kind[f] == 2) {
                any = true;
                if (p.rows[f] * (long)p.dim[f] * 4 <= (256L << 20)) nt = false;
            }
        nt = nt && any;
        if (nt_env >= 0) nt = nt_env != 0;
        const long groups = ((long)B * L_out + U - 1) / U;
        dim3 fgrid((unsigned)((groups * g + 255) / 256));
        hipStream_t fst = (hipStream_t)stream;
#define T4R_FAST_L(G, UU, NC)                                                                                                  \
    do {                                                                                                                       \
        if (nt) hipLaunchKernelGGL((seq_features_fwd_fast_kernel<G, UU, NC, true>), fgrid, dim3(256), 0, fst, p);               \
        else hipLaunchKernelGGL((seq_features_fwd_fast_kernel<G, UU, NC, false>), fgrid, dim3(256), 0, fst, p);                 \
    } while (0)
#define T4R_FAST(G)                                                                                          \
    do {                                                                                                     \
        if (U == 8) T4R_FAST_L(G, 8, 1);                                                                     \
        else if (U == 4) T4R_FAST_L(G, 4, 1);                                                                \
        else T4R_FAST_L(G, 2, 1);                                                                            \
    } while (0)
        if (nch > 1) {      // rows wider than 256 floats: the 64-lane group, 2 or 4 chunks per lane
            if (nch == 2 && U == 4) T4R_FAST_L(64, 4, 2);
            else if (nch == 2) T4R_FAST_L(64, 2, 2);
            else T4R_FAST_L(64, 2, 4);
        } else switch (g) {
            case 8: T4R_FAST(8); break;
            case 16: T4R_FAST(16); break;
            case 32: T4R_FAST(32); break;
            default: T4R_FAST(64); break;
        }
#undef T4R_FAST_L
#undef T4R_FAST
        T4R_LAUNCH_CHECK();
        return 0;
    }
    const long threads = (long)B * L_out * g;
    dim3 grid((unsigned)((threads + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    switch (g) {
        case 8: hipLaunchKernelGGL(seq_features_fwd_kernel<8>, grid, dim3(256), 0, st, p); break;
        case 16: hipLaunchKernelGGL(seq_features_fwd_kernel<16>, grid, dim3(256), 0, st, p); break;
        case 32: hipLaunchKernelGGL(seq_features_fwd_kernel<32>, grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL(seq_features_fwd_kernel<64>, grid, dim3(256), 0, st, p); break;
    }
    T4R_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// apply mask as its own pass (used after the projection MLP):  in place on x [B*L_out, H]
//   MLM: x = mask ? memb : x ;  CLM: x = mask ? (l==L-1 ? 0 : x) : memb ;  CLM infer: mask ? x : memb
// For MLM inference (L_out = L_in + 1) the caller gathers with L_out first.
__global__ __launch_bounds__(256) void apply_mask_kernel(float* __restrict__ x,
                                                          const unsigned char* __restrict__ mask,
                                                          const float* __restrict__ memb, long ntok,
                                                          int L, int H, int mode) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int hq = (H + 3) / 4;
    const long tok = i / hq;
    if (tok >= ntok) return;
    const int c0 = (int)(i % hq) * 4;
    const bool m = mask[tok] != 0;
    const int l = (int)(tok % L);
    bool keep, zero = false;
    if (mode == MASK_MLM) keep = !m;
    else if (mode == MASK_CLM) { keep = m; zero = (l == L - 1); }
    else keep = m;
    if (keep && !zero) return;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (c0 + e < H) x[tok * H + c0 + e] = keep ? 0.f : memb[c0 + e];
}

extern "C" int t4r_apply_mask_fwd(void* stream, float* x, const unsigned char* mask,
                                  const float* masked_emb, int B, int L, int H, int mode) {
    const long ntok = (long)B * L;
    if (ntok == 0 || mode == MASK_NONE) return 0;
    const long n = ntok * ((H + 3) / 4);
    hipLaunchKernelGGL(apply_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, x, mask, masked_emb, ntok, L, H, mode);
    T4R_LAUNCH_CHECK();
    return 0;
}

// backward of apply mask: d_memb[H] += sum over replaced tokens of dy ; dy <- 0 at replaced /
// zeroed tokens (in place), so that the upstream backward sees d x.
// The sum is formed in two stages (per-workgroup partial rows, then t4r_reduce_partials_launch in block order): 640
// workgroups adding into the same H addresses with atomics took 24 us and gave a different rounding every run.
#define T4R_MASK_BWD_TOK 32
// src == dst: in place (only the zeroed elements are written); else every element of dst is written (dst = src where kept)
__global__ __launch_bounds__(256) void apply_mask_bwd_kernel(const float* src, float* dy,
                                                              const unsigned char* __restrict__ mask,
                                                              float* __restrict__ part, long ntok,
                                                              int L, int H, int mode) {
    // 256 threads = column slots x token slices: with H = 128 two slices of 16 tokens each (a single slice walked its
    // tokens one after the other: latency bound)
    __shared__ float sh[256];
    const long t0 = (long)blockIdx.x * T4R_MASK_BWD_TOK;
    const long t1 = min(ntok, t0 + T4R_MASK_BWD_TOK);
    const int ncol = H < 256 ? H : 256;                   // column slots per pass
    const int nsl = 256 / ncol > 0 ? 256 / ncol : 1;      // token slices
    const int cs = threadIdx.x % ncol, sl = threadIdx.x / ncol;
    for (int c0 = 0; c0 < H; c0 += ncol) {
        const int c = c0 + cs;
        float acc = 0.f;
        if (c < H && sl < nsl) {
            for (long t = t0 + sl; t < t1; t += nsl) {
                const bool m = mask[t] != 0;
                const int l = (int)(t % L);
                bool keep, zero = false;
                if (mode == MASK_MLM) keep = !m;
                else if (mode == MASK_CLM) { keep = m; zero = (l == L - 1); }
                else keep = m;
                const float v = src[t * H + c];
                if (!keep) { acc += v; dy[t * H + c] = 0.f; }
                else if (zero) dy[t * H + c] = 0.f;
                else if (src != dy) dy[t * H + c] = v;
            }
        }
        sh[threadIdx.x] = acc;
        __syncthreads();
        if (sl == 0 && c < H) {
            for (int k = 1; k < nsl; ++k) acc += sh[k * ncol + cs];       // fixed order
            part[(long)blockIdx.x * H + c] = acc;
        }
        __syncthreads();
    }
}

// The same for H % 4 == 0, H <= 1024 (round 6): float4 columns, 256 / (H / 4) token rows of the block in parallel and four
// row requests in flight per thread (the scalar form above walks 16 tokens per thread one dependent load at a time: 29 us for
// the 10 MB of BASELINE configs[1] once it also writes every element).  Fixed summation order: a thread's tokens ascending,
// then the token slices in order.
__global__ __launch_bounds__(256) void apply_mask_bwd4_kernel(const float* src, float* dy,
                                                               const unsigned char* __restrict__ mask,
                                                               float* __restrict__ part, long ntok, int L, int H, int mode) {
    __shared__ float4 sh[256];
    const int hq = H >> 2, nsl = 256 / hq;
    const int cs = threadIdx.x % hq, sl = threadIdx.x / hq;
    const long t0 = (long)blockIdx.x * T4R_MASK_BWD_TOK;
    const long t1 = min(ntok, t0 + T4R_MASK_BWD_TOK);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sl < nsl) {
        for (long tb = t0 + sl; tb < t1; tb += 4L * nsl) {
            float4 v[4];
            unsigned char mk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {            // unconditional loads from clamped rows: four requests in flight
                const long tc = min(tb + (long)u * nsl, t1 - 1);
                v[u] = *reinterpret_cast<const float4*>(src + tc * H + 4 * cs);
                mk[u] = mask[tc];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long t = tb + (long)u * nsl;
                if (t >= t1) break;
                const bool m = mk[u] != 0;
                const int l = (int)(t % L);
                bool keep, zero = false;
                if (mode == MASK_MLM) keep = !m;
                else if (mode == MASK_CLM) { keep = m; zero = (l == L - 1); }
                else keep = m;
                float4* d = reinterpret_cast<float4*>(dy + t * H + 4 * cs);
                if (!keep) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; *d = make_float4(0.f, 0.f, 0.f, 0.f); }
                else if (zero) *d = make_float4(0.f, 0.f, 0.f, 0.f);
                else if (src != dy) *d = v[u];
            }
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (sl == 0) {
        for (int k = 1; k < nsl; ++k) {              // fixed order
            const float4 o = sh[k * hq + cs];
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        *reinterpret_cast<float4*>(part + (long)blockIdx.x * H + 4 * cs) = acc;
    }
}
static void apply_mask_bwd_launch(hipStream_t st, const float* src, float* dst, const unsigned char* mask, float* ws, long ntok,
                                  int L, int H, int mode, int nblk) {
    const bool vec = H % 4 == 0 && H <= 1024 && (((uintptr_t)src | (uintptr_t)dst | (uintptr_t)ws) & 15) == 0;
    if (vec) hipLaunchKernelGGL(apply_mask_bwd4_kernel, dim3(nblk), dim3(256), 0, st, src, dst, mask, ws, ntok, L, H, mode);
    else hipLaunchKernelGGL(apply_mask_bwd_kernel, dim3(nblk), dim3(256), 0, st, src, dst, mask, ws, ntok, L, H, mode);
}

int t4r_reduce_partials_launch(hipStream_t st, const float* part, int nblocks, float* o0, int n0, int a0,
                               float* o1, int n1, int a1, float* o2, int n2, int a2);   // elementwise.hip
extern "C" long t4r_apply_mask_bwd_ws_floats(int B, int L, int H) {
    return (((long)B * L + T4R_MASK_BWD_TOK - 1) / T4R_MASK_BWD_TOK) * H;
}
// ws: t4r_apply_mask_bwd_ws_floats(B, L, H) floats of scratch
extern "C" int t4r_apply_mask_bwd(void* stream, float* dy, const unsigned char* mask, float* d_memb,
                                  int B, int L, int H, int mode, float* ws) {
    const long ntok = (long)B * L;
    if (ntok == 0 || mode == MASK_NONE) return 0;
    T4R_CHECK_ARG(ws, "apply_mask_bwd: null workspace");
    const int nblk = (int)((ntok + T4R_MASK_BWD_TOK - 1) / T4R_MASK_BWD_TOK);
    apply_mask_bwd_launch((hipStream_t)stream, dy, dy, mask, ws, ntok, L, H, mode, nblk);
    T4R_LAUNCH_CHECK();
    return t4r_reduce_partials_launch((hipStream_t)stream, ws, nblk, d_memb, H, 1, nullptr, 0, 0, nullptr, 0, 0);
}
// the same out of place: dx = the masked gradient, dy untouched (an autograd backward must not write its incoming gradient:
// the module path cloned dy first -- a 10 MB device copy per step at BASELINE configs[1] -- and masked the clone in place)
extern "C" int t4r_apply_mask_bwd_to(void* stream, const float* dy, float* dx, const unsigned char* mask, float* d_memb,
                                     int B, int L, int H, int mode, float* ws) {
    const long ntok = (long)B * L;
    if (ntok == 0) return 0;
    T4R_CHECK_ARG(dy && dx && dy != dx, "apply_mask_bwd_to: dy and dx must be distinct buffers");
    if (mode == MASK_NONE) {
        return (int)hipMemcpyAsync(dx, dy, sizeof(float) * ntok * H, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    }
    T4R_CHECK_ARG(ws, "apply_mask_bwd_to: null workspace");
    const int nblk = (int)((ntok + T4R_MASK_BWD_TOK - 1) / T4R_MASK_BWD_TOK);
    apply_mask_bwd_launch((hipStream_t)stream, dy, dx, mask, ws, ntok, L, H, mode, nblk);
    T4R_LAUNCH_CHECK();
    return t4r_reduce_partials_launch((hipStream_t)stream, ws, nblk, d_memb, H, 1, nullptr, 0, 0, nullptr, 0, 0);
}

// ------------------------------------------------------------------------------------------
// backward of the gather: for categorical feature f, d_table[id, :] += d_out[tok, col:col+dim]
// for id != padding_idx (nn.Embedding(padding_idx=0): row 0 gets no lookup gradient).
// For element-wise-sum aggregation every feature receives d_out itself.
// MLM inference duplication is forward-only, so L_in == L_out here.
// Small tables (rows*dim*4 <= 48 KiB) are accumulated in LDS per block first: a 10-row table
// hit by 20k tokens would otherwise serialise ~2k atomics per address.
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const float* __restrict__ dout,
                                                             const long* __restrict__ ids,
                                                             float* __restrict__ dtable, long ntok,
                                                             int W, int col, int dim, long rows,
                                                             int padding_idx, int tok_per_block,
                                                             int use_lds, int ids_div) {
    extern __shared__ float lds[];
    const long t0 = (long)blockIdx.x * tok_per_block;
    const long t1 = min(ntok, t0 + tok_per_block);
    if (use_lds) {
        const int n = (int)(rows * dim);
        for (int i = threadIdx.x; i < n; i += 256) lds[i] = 0.f;
        __syncthreads();
        for (long t = t0 + (threadIdx.x / 64); t < t1; t += 4) {     // one token per wave, lanes over columns
            const long id = ids[t / ids_div];
            if (id == padding_idx || id < 0 || id >= rows) continue;
            for (int c = threadIdx.x & 63; c < dim; c += 64)
                atomicAdd(&lds[id * dim + c], dout[t * W + col + c]);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256)
            if (lds[i] != 0.f) atomicAdd(dtable + i, lds[i]);
    } else {
        const int dq = dim;  // one lane per column, tokens striped over waves
        for (long t = t0 + (threadIdx.x / 64); t < t1; t += 4) {
            const long id = ids[t / ids_div];
            if (id == padding_idx || id < 0 || id >= rows) continue;
            for (int c = threadIdx.x & 63; c < dq; c += 64)
                atomicAdd(dtable + id * dim + c, dout[t * W + col + c]);
        }
    }
}

// ids_div: 1 for sequence features (one id per token), L for per-session features (id index = tok / L)
extern "C" int t4r_embedding_bwd(void* stream, const float* dout, const long* ids, float* dtable,
                                 long ntok, int W, int col, int dim, long rows, int padding_idx,
                                 int ids_div) {
    if (ntok == 0) return 0;
    T4R_CHECK_ARG(ids_div >= 1, "embedding_bwd: ids_div >= 1");
    const int use_lds = rows * dim * 4 <= 48 * 1024;
    const int tpb = use_lds ? 64 : 32;
    const size_t smem = use_lds ? (size_t)rows * dim * 4 : 0;
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3((unsigned)((ntok + tpb - 1) / tpb)), dim3(256), smem,
                       (hipStream_t)stream, dout, ids, dtable, ntok, W, col, dim, rows, padding_idx,
                       tpb, use_lds, ids_div);
    T4R_LAUNCH_CHECK();
    return 0;
}

// item-multi aggregation backward helper: out = item * other  =>
//   d_item = d_out * other ; d_other = d_out * item.  The host recomputes `other`/`item` rows
// with t4r_seq_features_fwd (AGG_SUM over the respective feature subsets) and multiplies here.
__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a,
                                                   const float* __restrict__ b,
                                                   float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}
extern "C" int t4r_mul(void* stream, const float* a, const float* b, float* out, long n) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, a, b, out, n);
    T4R_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// SoftEmbedding (+ per-feature LayerNorm) forward for one continuous feature:
//   s_k = x*pw_k + pb_k ; w = softmax(s) ; e = sum_k w_k T[k,:] ; y = LN(e)*g + b  (g null: y = e)
// One thread per token, K/D bounded by template maxima so everything stays in registers.
// EXACT: K == KMAX and D == DMAX are compile-time facts (the reference's defaults, 10 soft bins x 8 dimensions, get their own
// instantiation): with run-time K, D every `k < K` / `d < D` of the unrolled loops is a scalar branch and every table element a
// scalar load waited for behind its branch -- 9 k instructions, ~600 branches, 306 dependent s_load in the backward, 58 us per
// launch at BASELINE configs[2]; the exact form is straight-line code with its parameter loads batched up front.
template <int KMAX, int DMAX, bool EXACT = false>
__global__ __launch_bounds__(256) void soft_embedding_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ pw, const float* __restrict__ pb,
    const float* __restrict__ table, const float* __restrict__ lnw, const float* __restrict__ lnb,
    float* __restrict__ out, long ntok, int K_rt, int D_rt, float eps) {
    const int K = EXACT ? KMAX : K_rt, D = EXACT ? DMAX : D_rt;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntok) return;
    const float xv = x[t];
    float w[KMAX], e[DMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { w[k] = k < K ? xv * pw[k] + pb[k] : -INFINITY; mx = fmaxf(mx, w[k]); }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { w[k] = k < K ? expf(w[k] - mx) : 0.f; den += w[k]; }
    const float inv = 1.f / den;
#pragma unroll
    for (int d = 0; d < DMAX; ++d) {
        float a = 0.f;
        if (d < D) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) a += (w[k] * inv) * table[k * D + d];
        }
        e[d] = a;
    }
    if (lnw) {
        float mu = 0.f;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) mu += e[d];
        mu /= D;
        float var = 0.f;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) if (d < D) var += (e[d] - mu) * (e[d] - mu);
        const float rs = rsqrtf(var / D + eps);
#pragma unroll
        for (int d = 0; d < DMAX; ++d) if (d < D) e[d] = (e[d] - mu) * rs * lnw[d] + lnb[d];
    }
#pragma unroll
    for (int d = 0; d < DMAX; ++d) if (d < D) out[t * D + d] = e[d];
}

// T4R_SOFT_EXACT=0: the general (run-time K, D) kernels for every shape (A/B and parity of the exact-shape instantiation)
static bool soft_exact_on() {
    static int on = -1;
    if (on < 0) { const char* e = t4r_exp_getenv("T4R_SOFT_EXACT"); on = e ? (atoi(e) != 0) : 1; }
    return on != 0;
}

extern "C" int t4r_soft_embedding_fwd(void* stream, const float* x, const float* proj_w,
                                      const float* proj_b, const float* table, const float* ln_w,
                                      const float* ln_b, float* out, long ntok, int K, int D,
                                      float eps) {
    if (ntok == 0) return 0;
    T4R_CHECK_ARG(K >= 1 && K <= T4R_SOFT_MAXK && D >= 1 && D <= T4R_SOFT_MAXD,
                  "soft_embedding_fwd: K<=32, dim<=32");
    dim3 grid((unsigned)((ntok + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (K == 10 && D == 8 && soft_exact_on())
        hipLaunchKernelGGL((soft_embedding_fwd_kernel<10, 8, true>), grid, block, 0, st, x, proj_w, proj_b,
                           table, ln_w, ln_b, out, ntok, K, D, eps);
    else if (K <= 16 && D <= 8)
        hipLaunchKernelGGL((soft_embedding_fwd_kernel<16, 8>), grid, block, 0, st, x, proj_w, proj_b,
                           table, ln_w, ln_b, out, ntok, K, D, eps);
    else
        hipLaunchKernelGGL((soft_embedding_fwd_kernel<32, 32>), grid, block, 0, st, x, proj_w, proj_b,
                           table, ln_w, ln_b, out, ntok, K, D, eps);
    T4R_LAUNCH_CHECK();
    return 0;
}

// SoftEmbedding + LayerNorm backward.  dy = d out [tok, W] (columns col..col+D):
//   d g, d b, d T, d pw, d pb   (x has no gradient).  One thread per token.  Every parameter gradient is a sum over ALL
// tokens of a per-token term: K D + 2 K + 2 D sums (116 at the defaults K = 10, D = 8).  Round 5: each sum is reduced
//   * over the wave with DPP row shifts + v_readlane (the shuffle-based wave_sum is a chain of six ds_bpermute round
//     trips per sum: 58 us per launch at C3, the kernel was nothing but that chain),
//   * over the four waves of the workgroup through one LDS slot per wave, added in wave order,
//   * over the workgroups by per-workgroup partial rows + t4r_reduce_partials_launch in block order
// -- no atomics anywhere: the soft-embedding gradients are bit-reproducible like every other gradient of the path.
__device__ __forceinline__ float wave_total_dpp(float v) {
    // after the four row shifts lane 15 of every row of 16 holds its row's sum (lanes without a source add 0)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, false));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 15));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 47));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    return (r0 + r1) + (r2 + r3);
}

template <int KMAX, int DMAX, bool EXACT = false>
__global__ __launch_bounds__(256) void soft_embedding_bwd_kernel(
    const float* __restrict__ dout, const float* __restrict__ x, const float* __restrict__ pw,
    const float* __restrict__ pb, const float* __restrict__ table, const float* __restrict__ lnw,
    float* __restrict__ partA, float* __restrict__ partB, long ntok, int W, int col, int K_rt, int D_rt,
    float eps) {
    const int K = EXACT ? KMAX : K_rt, D = EXACT ? DMAX : D_rt;      // EXACT: see soft_embedding_fwd_kernel
    // per-wave slots: [4][K D | K | K | D | D]
    __shared__ float red[4][KMAX * DMAX + 2 * KMAX + 2 * DMAX];
    const int nA = K * D + 2 * K, nB = 2 * D;
    const int wv = threadIdx.x >> 6;
    float* r_tab = red[wv];             // [K*D]
    float* r_pw = r_tab + K * D;        // [K]
    float* r_pb = r_pw + K;             // [K]
    float* r_g = r_pb + K;              // [D]
    float* r_b = r_g + D;               // [D]
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t < ntok;
    const int lane = threadIdx.x & 63;
    auto wput = [&](float* dst, float v) {          // this wave's sum -> its slot (every slot is written exactly once)
        v = wave_total_dpp(live ? v : 0.f);
        if (lane == 0) *dst = v;
    };
    {
        const float xv = live ? x[t] : 0.f;
        float w[KMAX], e[DMAX], de[DMAX];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { w[k] = k < K ? xv * pw[k] + pb[k] : -INFINITY; mx = fmaxf(mx, w[k]); }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) { w[k] = k < K ? expf(w[k] - mx) : 0.f; den += w[k]; }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) w[k] /= den;
#pragma unroll
        for (int d = 0; d < DMAX; ++d) {
            float a = 0.f;
            if (d < D) {
#pragma unroll
                for (int k = 0; k < KMAX; ++k) if (k < K) a += w[k] * table[k * D + d];
            }
            e[d] = a;
        }
        const float* dy = dout + (live ? t : 0) * W + col;
        if (lnw) {
            float mu = 0.f;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) mu += e[d];
            mu /= D;
            float var = 0.f;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) if (d < D) var += (e[d] - mu) * (e[d] - mu);
            const float rs = rsqrtf(var / D + eps);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) if (d < D) {
                const float xh = (e[d] - mu) * rs;
                const float g = dy[d] * lnw[d];
                s1 += g; s2 += g * xh;
                wput(&r_g[d], dy[d] * xh);
                wput(&r_b[d], dy[d]);
            }
            s1 /= D; s2 /= D;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
                const float xh = (e[d] - mu) * rs;
                de[d] = d < D ? rs * (dy[d] * lnw[d] - s1 - xh * s2) : 0.f;
            }
        } else {
#pragma unroll
            for (int d = 0; d < DMAX; ++d) {
                de[d] = d < D ? dy[d] : 0.f;
                if (d < D && lane == 0) { r_g[d] = 0.f; r_b[d] = 0.f; }
            }
        }
        // e = sum_k w_k T_k :  dT_k += w_k de ; dw_k = de . T_k ; ds = w * (dw - sum_j w_j dw_j)
        float dw[KMAX];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            float a = 0.f;
            if (k < K) {
#pragma unroll
                for (int d = 0; d < DMAX; ++d) if (d < D) {
                    a += de[d] * table[k * D + d];
                    wput(&r_tab[k * D + d], w[k] * de[d]);
                }
            }
            dw[k] = a;
            dot += w[k] * a;
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) if (k < K) {
            const float ds = w[k] * (dw[k] - dot);
            wput(&r_pw[k], ds * xv);
            wput(&r_pb[k], ds);
        }
    }
    __syncthreads();
    // the four waves' slots in wave order -> this workgroup's partial rows
    for (int i = threadIdx.x; i < nA; i += 256)
        partA[(long)blockIdx.x * nA + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    for (int i = threadIdx.x; i < nB; i += 256)
        partB[(long)blockIdx.x * nB + i] = (red[0][nA + i] + red[1][nA + i]) + (red[2][nA + i] + red[3][nA + i]);
}

extern "C" long t4r_soft_embedding_bwd_ws_floats(long ntok, int K, int D) {
    return ((ntok + 255) / 256) * (long)(K * D + 2 * K + 2 * D);
}

// ws: t4r_soft_embedding_bwd_ws_floats(ntok, K, D) floats of scratch (per-workgroup partial sums)
int t4r_reduce_partials_launch(hipStream_t st, const float* part, int nblocks, float* o0, int n0, int a0,
                               float* o1, int n1, int a1, float* o2, int n2, int a2);   // elementwise.hip
extern "C" int t4r_soft_embedding_bwd(void* stream, const float* dout, const float* x,
                                      const float* proj_w, const float* proj_b, const float* table,
                                      const float* ln_w, float* d_proj_w, float* d_proj_b,
                                      float* d_table, float* d_ln_w, float* d_ln_b, long ntok, int W,
                                      int col, int K, int D, float eps, float* ws) {
    if (ntok == 0) return 0;
    T4R_CHECK_ARG(K >= 1 && K <= T4R_SOFT_MAXK && D >= 1 && D <= T4R_SOFT_MAXD,
                  "soft_embedding_bwd: K<=32, dim<=32");
    T4R_CHECK_ARG(ws != nullptr, "soft_embedding_bwd: workspace required (t4r_soft_embedding_bwd_ws_floats)");
    const int nblocks = (int)((ntok + 255) / 256);
    const int nA = K * D + 2 * K, nB = 2 * D;
    float* partA = ws;
    float* partB = ws + (long)nblocks * nA;
    dim3 grid((unsigned)nblocks), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (K == 10 && D == 8 && soft_exact_on())
        hipLaunchKernelGGL((soft_embedding_bwd_kernel<10, 8, true>), grid, block, 0, st, dout, x, proj_w,
                           proj_b, table, ln_w, partA, partB, ntok, W, col, K, D, eps);
    else if (K <= 16 && D <= 8)
        hipLaunchKernelGGL((soft_embedding_bwd_kernel<16, 8>), grid, block, 0, st, dout, x, proj_w,
                           proj_b, table, ln_w, partA, partB, ntok, W, col, K, D, eps);
    else
        hipLaunchKernelGGL((soft_embedding_bwd_kernel<32, 32>), grid, block, 0, st, dout, x, proj_w,
                           proj_b, table, ln_w, partA, partB, ntok, W, col, K, D, eps);
    T4R_LAUNCH_CHECK();
    // fixed-order sums over the workgroups, accumulated into the gradients
    int rc = t4r_reduce_partials_launch(st, partA, nblocks, d_table, K * D, 1, d_proj_w, K, 1, d_proj_b, K, 1);
    if (rc != 0) return rc;
    if (ln_w) rc = t4r_reduce_partials_launch(st, partB, nblocks, d_ln_w, D, 1, d_ln_b, D, 1, nullptr, 0, 0);
    return rc;
}

// ------------------------------------------------------------------------------------------
// ragged -> padded  (pad_inputs / _pad_ragged_tensor): out[r, c] = values[offsets[r] + c] for
// c < min(len_r, L), else 0.  One wave per row; elem_size 4 or 8 bytes (float / int64).
template <typename T>
__global__ __launch_bounds__(256) void ragged_to_padded_kernel(const T* __restrict__ values,
                                                                const long* __restrict__ offsets,
                                                                T* __restrict__ out, int rows, int L) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long o0 = offsets[row], o1 = offsets[row + 1];
    const long len = o1 - o0;
    for (int c = threadIdx.x & 63; c < L; c += 64)
        out[(long)row * L + c] = c < len ? values[o0 + c] : (T)0;
}

// same, for a batch given by row ids into a device-resident ragged column (shuffled batches of the
// device-resident feed): out[i, c] = values[offsets[row_ids[i]] + c].  Scalar (per-session) columns
// are the L = 1 case with offsets == nullptr: out[i] = values[row_ids[i]].
template <typename T>
__global__ __launch_bounds__(256) void ragged_gather_to_padded_kernel(const T* __restrict__ values,
                                                                       const long* __restrict__ offsets,
                                                                       const long* __restrict__ row_ids,
                                                                       T* __restrict__ out, int rows, int L) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= rows) return;
    const long row = row_ids[i];
    const long o0 = offsets ? offsets[row] : row;
    const long len = offsets ? offsets[row + 1] - o0 : 1;
    for (int c = threadIdx.x & 63; c < L; c += 64)
        out[(long)i * L + c] = c < len ? values[o0 + c] : (T)0;
}

extern "C" int t4r_ragged_gather_to_padded(void* stream, const void* values, const long* offsets,
                                           const long* row_ids, void* out, int rows, int L, int elem_size) {
    if (rows == 0 || L == 0) return 0;
    T4R_CHECK_ARG(elem_size == 4 || elem_size == 8, "ragged_gather_to_padded: elem_size 4 or 8");
    T4R_CHECK_ARG(values && row_ids && out, "ragged_gather_to_padded: null pointer");
    T4R_CHECK_ARG(offsets || L == 1, "ragged_gather_to_padded: a scalar column has L = 1");
    dim3 grid((rows + 3) / 4), block(256);
    if (elem_size == 8)
        hipLaunchKernelGGL(ragged_gather_to_padded_kernel<long>, grid, block, 0, (hipStream_t)stream,
                           (const long*)values, offsets, row_ids, (long*)out, rows, L);
    else
        hipLaunchKernelGGL(ragged_gather_to_padded_kernel<float>, grid, block, 0, (hipStream_t)stream,
                           (const float*)values, offsets, row_ids, (float*)out, rows, L);
    T4R_LAUNCH_CHECK();
    return 0;
}

// row-length maximum (pad_inputs: min(max_sequence_length, batch max)); result in *out_max
__global__ void max_row_len_kernel(const long* __restrict__ offsets, int rows, int* out_max) {
    int m = 0;
    for (int r = threadIdx.x; r < rows; r += blockDim.x) m = max(m, (int)(offsets[r + 1] - offsets[r]));
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    __shared__ int sm[16];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = max(m, sm[i]);
        *out_max = m;
    }
}

extern "C" int t4r_ragged_max_len(void* stream, const long* offsets, int rows, int* out_max) {
    T4R_CHECK_ARG(rows >= 1, "ragged_max_len: rows >= 1");
    hipLaunchKernelGGL(max_row_len_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, offsets, rows, out_max);
    T4R_LAUNCH_CHECK();
    return 0;
}

extern "C" int t4r_ragged_to_padded(void* stream, const void* values, const long* offsets, void* out,
                                    int rows, int L, int elem_size) {
    if (rows == 0 || L == 0) return 0;
    T4R_CHECK_ARG(elem_size == 4 || elem_size == 8, "ragged_to_padded: elem_size 4 or 8");
    dim3 grid((rows + 3) / 4), block(256);
    if (elem_size == 8)
        hipLaunchKernelGGL(ragged_to_padded_kernel<long>, grid, block, 0, (hipStream_t)stream,
                           (const long*)values, offsets, (long*)out, rows, L);
    else
        hipLaunchKernelGGL(ragged_to_padded_kernel<float>, grid, block, 0, (hipStream_t)stream,
                           (const float*)values, offsets, (float*)out, rows, L);
    T4R_LAUNCH_CHECK();
    return 0;
}
