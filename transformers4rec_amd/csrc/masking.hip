// Masking-schema / label generation for the session-sequence path (integer kernels; results
// must be BIT-EXACT against the reference) plus the ordered label-row compaction that feeds
// the next-item head.
//
// Reference behaviour restated (transformers4rec/torch/masking.py):
//   MaskedLanguageModeling._compute_masked_targets   :376-470
//       train     :425-459  bernoulli(p) & non_pad -> force one label (multinomial over non-pad)
//                           -> if every non-pad item is a label, un-label one (multinomial over labels)
//       eval      :460-468  last item only | predict_all
//       inference :406-418  [B, L+1] with the [MASK] slot at index len
//   CausalLanguageModeling._compute_masked_targets   :274-300   labels = ids shifted left
//   MaskSequence.predict_all                         :182-213
//   NextItemPredictionTask label-row selection       transformers4rec/torch/model/prediction_task.py:436-443
//       rows where masked_targets.flatten() != padding_idx, row-major (b, l) order.
//
// The training path has two sources for its random draws:
//   * injected draws (bern[B,L], j1[B], j2[B]) -- what the parity tests replay from the
//     reference's recorded torch.bernoulli / torch.multinomial calls;
//   * device Philox draws (seed, offset) -- the production path: j1 uniform over the non-pad
//     positions, j2 uniform over the labelled positions (the distributions torch.multinomial
//     samples from with 0/1 weights).
// One wave per session row; a row's non-pad / label sets are 64-bit ballots (L <= 64*MAXC - 1; MAXC is a template parameter:
// 4 words up to 255 positions -- every configuration of BASELINE.json --, 16 up to 1023).
#include "t4r_common.h"


enum { MLM_TRAIN = 0, MLM_EVAL_LAST = 1, MLM_EVAL_ALL = 2, MLM_INFER = 3,
       CLM_TRAIN = 4, CLM_LAST = 5, CLM_INFER = 6 };

// index of the k-th (0-based) set bit over the chunked bitset; -1 if none
template <int MAXC>
__device__ __forceinline__ int select_kth(const unsigned long long (&bits)[MAXC + 1], int k) {
    int res = -1;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int n = __popcll(bits[c]);
        if (res < 0 && k >= 0 && k < n) {
            unsigned long long b = bits[c];
            for (int i = 0; i < k; ++i) b &= b - 1;  // clear k lowest set bits
            res = c * 64 + __ffsll((long long)b) - 1;
        }
        k -= n;
    }
    return res;
}
// static-index helpers (runtime-indexed register arrays would spill to scratch)
template <int MAXC>
__device__ __forceinline__ void bit_or(unsigned long long (&bits)[MAXC + 1], int idx, const unsigned long long (&andmask)[MAXC + 1], bool use_and) {
#pragma unroll
    for (int c = 0; c < MAXC + 1; ++c)
        if (c == (idx >> 6)) bits[c] |= (1ull << (idx & 63)) & (use_and ? andmask[c] : ~0ull);
}
template <int MAXC>
__device__ __forceinline__ void bit_clear(unsigned long long (&bits)[MAXC + 1], int idx) {
#pragma unroll
    for (int c = 0; c < MAXC + 1; ++c)
        if (c == (idx >> 6)) bits[c] &= ~(1ull << (idx & 63));
}
template <int MAXC>
__device__ __forceinline__ int popc_all(const unsigned long long (&bits)[MAXC + 1]) {
    int n = 0;
#pragma unroll
    for (int c = 0; c < MAXC + 1; ++c) n += __popcll(bits[c]);
    return n;
}

template <int MAXC>
__global__ __launch_bounds__(256) void mask_targets_kernel(
    const long* __restrict__ ids, int B, int L, int mode, long padding_idx,
    // injected draws (train) -- any may be null => Philox
    const unsigned char* __restrict__ bern, const long* __restrict__ j1_in,
    const long* __restrict__ j2_in, float p, unsigned long long seed, unsigned long long offset,
    unsigned char* __restrict__ mask_out, long* __restrict__ labels_out, int* __restrict__ row_count) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    const int Lout = (mode == MLM_INFER) ? L + 1 : L;
    const int nchunk = (L + 63) / 64;
    const int nchunk_out = (Lout + 63) / 64;
    const long* row = ids + (long)b * L;

    unsigned long long nonpad[MAXC + 1], lab[MAXC + 1];
    long myid[MAXC + 1];
#pragma unroll
    for (int c = 0; c < MAXC + 1; ++c) { nonpad[c] = 0; lab[c] = 0; myid[c] = padding_idx; }
    int n_nonpad = 0;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < nchunk) {
            const int l = c * 64 + lane;
            myid[c] = l < L ? row[l] : padding_idx;
            nonpad[c] = __ballot(myid[c] != padding_idx);
            n_nonpad += __popcll(nonpad[c]);
        }
    }
    // label value per lane/chunk (what labels_out gets where lab bit is set)
    long labval[MAXC + 1];
#pragma unroll
    for (int c = 0; c < MAXC + 1; ++c) labval[c] = myid[c < MAXC ? c : MAXC - 1];
    // mask schema may differ from (labels != pad) in the CLM last-item variants
    bool mask_is_nonpad = false;

    if (mode == MLM_TRAIN) {
        Philox rng(seed);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            if (c < nchunk) {
                const int l = c * 64 + lane;
                bool hit;
                if (bern) hit = l < L && bern[(long)b * L + l] != 0;
                else {
                    const uint4 r = rng(offset + (unsigned long long)b * L + l, 0);
                    hit = l < L && u32_to_unit(r.x) < p;
                }
                lab[c] = __ballot(hit) & nonpad[c];
            }
        }
        if (n_nonpad > 0) {
            int j1, j2;
            uint4 r = rng(offset + (unsigned long long)b, 1);
            if (j1_in) j1 = (int)j1_in[b];
            else j1 = select_kth<MAXC>(nonpad, min(n_nonpad - 1, (int)(u32_to_unit(r.x) * n_nonpad)));
            if (j1 >= 0 && j1 < L) bit_or<MAXC>(lab, j1, nonpad, true);
            const int n_lab = popc_all<MAXC>(lab);
            if (n_lab == n_nonpad) {
                if (j2_in) j2 = (int)j2_in[b];
                else j2 = select_kth<MAXC>(lab, min(n_lab - 1, (int)(u32_to_unit(r.y) * n_lab)));
                if (j2 >= 0 && j2 < L) bit_clear<MAXC>(lab, j2);
            }
        }
    } else if (mode == MLM_EVAL_LAST) {
        // labels[b, len-1] = ids[b, len-1]   (len = count of non-pad; reference indexes by count)
        const int last = n_nonpad - 1;  // -1 wraps to L-1 in torch indexing
        const int idx = last < 0 ? L - 1 : last;
        bit_or<MAXC>(lab, idx, nonpad, true);
    } else if (mode == MLM_EVAL_ALL || mode == CLM_TRAIN || mode == CLM_LAST) {
        // predict_all: labels[l] = ids[l+1] (0 at L-1); mask = labels != pad
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            if (c < nchunk) {
                const int l = c * 64 + lane;
                const long nxt = (l + 1 < L) ? row[l + 1] : padding_idx;
                labval[c] = nxt;
                lab[c] = __ballot(l < L && nxt != padding_idx);
            }
        }
        if (mode == CLM_LAST) {
            const int last = popc_all<MAXC>(lab) - 1;
            const int idx = last < 0 ? L - 1 : last;
            // keep only labels[idx] (it may itself be pad -> no label at all)
#pragma unroll
            for (int c = 0; c < MAXC + 1; ++c) lab[c] &= (c == (idx >> 6)) ? (1ull << (idx & 63)) : 0ull;
            mask_is_nonpad = true;
        }
    } else if (mode == MLM_INFER) {
        // labels[b, len] = ids[b, len-1]  on a [B, L+1] grid
        const int idx = n_nonpad;                 // 0..L
        const int src = n_nonpad - 1 < 0 ? L - 1 : n_nonpad - 1;
        const long v = row[src];
        if (v != padding_idx) bit_or<MAXC>(lab, idx, nonpad, false);
#pragma unroll
        for (int c = 0; c < MAXC + 1; ++c) labval[c] = v;
    } else {  // CLM_INFER: mask = ids != pad, labels = ids
#pragma unroll
        for (int c = 0; c < MAXC; ++c) lab[c] = nonpad[c];
    }

    int cnt = 0;
#pragma unroll
    for (int c = 0; c < MAXC + 1; ++c) {
        if (c < nchunk_out) {
            const int l = c * 64 + lane;
            if (l < Lout) {
                const bool is_lab = (lab[c] >> lane) & 1ull;
                const bool np = c < MAXC ? ((nonpad[c] >> lane) & 1ull) : false;
                mask_out[(long)b * Lout + l] = mask_is_nonpad ? np : is_lab;
                labels_out[(long)b * Lout + l] = is_lab ? labval[c] : padding_idx;
            }
            cnt += __popcll(lab[c]);
        }
    }
    if (lane == 0 && row_count) row_count[b] = cnt;
}

extern "C" int t4r_mask_targets(void* stream, const long* item_ids, int B, int L, int mode,
                                long padding_idx, const unsigned char* bern, const long* j1,
                                const long* j2, float mlm_probability, unsigned long long seed,
                                unsigned long long offset, unsigned char* mask_schema,
                                long* masked_targets, int* row_count) {
    T4R_CHECK_ARG(L >= 1 && L <= 1023, "mask_targets: L must be in [1, 1023]");
    T4R_CHECK_ARG(mode >= 0 && mode <= CLM_INFER, "mask_targets: unknown mode");
    if (B == 0) return 0;
    if (L <= 255)
        hipLaunchKernelGGL(mask_targets_kernel<4>, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                           item_ids, B, L, mode, padding_idx, bern, j1, j2, mlm_probability, seed, offset,
                           mask_schema, masked_targets, row_count);
    else
        hipLaunchKernelGGL(mask_targets_kernel<16>, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                           item_ids, B, L, mode, padding_idx, bern, j1, j2, mlm_probability, seed, offset,
                           mask_schema, masked_targets, row_count);
    T4R_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// ordered compaction of the label positions (remove_pad_3d order: row-major over (b, l)).
//   step 1: exclusive scan of row_count[B]   -> row_offset[B], n_labels
//   step 2: one wave per row writes label_pos[row_offset[b] + rank] = b*L + l and the label.
__global__ __launch_bounds__(1024) void scan_rows_kernel(const int* __restrict__ row_count,
                                                          int* __restrict__ row_offset, int B,
                                                          int* __restrict__ n_total) {
    __shared__ int wsum[16];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < B; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < B ? row_count[i] : 0;
        int x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if ((threadIdx.x & 63) >= o) x += y;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
        const int incl = carry + woff + x;
        if (i < B) row_offset[i] = incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_total = carry;
}

__global__ __launch_bounds__(256) void compact_labels_kernel(const long* __restrict__ labels,
                                                              const int* __restrict__ row_offset,
                                                              int B, int L, long padding_idx,
                                                              int* __restrict__ label_pos,
                                                              long* __restrict__ labels_compact) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    int base = row_offset[b];
    for (int l0 = 0; l0 < L; l0 += 64) {
        const int l = l0 + lane;
        const long v = l < L ? labels[(long)b * L + l] : padding_idx;
        const bool keep = v != padding_idx;
        const unsigned long long bal = __ballot(keep);
        if (keep) {
            const int rank = __popcll(bal & ((1ull << lane) - 1ull));
            label_pos[base + rank] = b * L + l;
            labels_compact[base + rank] = v;
        }
        base += __popcll(bal);
    }
}

extern "C" int t4r_compact_labels(void* stream, const long* masked_targets, const int* row_count,
                                  int B, int L, long padding_idx, int* row_offset, int* n_labels,
                                  int* label_pos, long* labels_compact) {
    if (B == 0) { return 0; }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(scan_rows_kernel, dim3(1), dim3(1024), 0, st, row_count, row_offset, B, n_labels);
    hipLaunchKernelGGL(compact_labels_kernel, dim3((B + 3) / 4), dim3(256), 0, st, masked_targets,
                       row_offset, B, L, padding_idx, label_pos, labels_compact);
    T4R_LAUNCH_CHECK();
    return 0;
}

// gather rows: out[n, :] = x[pos[n], :]      (remove_pad_3d / inference last-position gather)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x,
                                                           const int* __restrict__ pos,
                                                           float* __restrict__ out, int n, int D) {
    const int dq = D / 4;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n * dq) return;
    const int r = (int)(i / dq), c = (int)(i % dq) * 4;
    *reinterpret_cast<float4*>(out + (long)r * D + c) =
        *reinterpret_cast<const float4*>(x + (long)pos[r] * D + c);
}
// scatter rows (backward): dx[pos[n], :] += dout[n, :]   (positions are unique)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ dout,
                                                            const int* __restrict__ pos,
                                                            float* __restrict__ dx, int n, int D) {
    const int dq = D / 4;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n * dq) return;
    const int r = (int)(i / dq), c = (int)(i % dq) * 4;
    float4* d = reinterpret_cast<float4*>(dx + (long)pos[r] * D + c);
    const float4 g = *reinterpret_cast<const float4*>(dout + (long)r * D + c);
    float4 o = *d;
    o.x += g.x; o.y += g.y; o.z += g.z; o.w += g.w;
    *d = o;
}

extern "C" int t4r_gather_rows(void* stream, const float* x, const int* pos, float* out, int n, int D) {
    if (n == 0) return 0;
    T4R_CHECK_ARG(D % 4 == 0, "gather_rows: D must be a multiple of 4");
    const long t = (long)n * (D / 4);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, x, pos, out, n, D);
    T4R_LAUNCH_CHECK();
    return 0;
}
extern "C" int t4r_scatter_rows_add(void* stream, const float* dout, const int* pos, float* dx, int n, int D) {
    if (n == 0) return 0;
    T4R_CHECK_ARG(D % 4 == 0, "scatter_rows: D must be a multiple of 4");
    const long t = (long)n * (D / 4);
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, dout, pos, dx, n, D);
    T4R_LAUNCH_CHECK();
    return 0;
}

// The head's backward as ONE launch (round 6): dx [T, D] = 0 except dx[pos[r], :] = scale * src[r, :] for the n label rows.
// Replaces torch.zeros (a 10 MB fill) + an element-wise multiply by the upstream gradient + scatter_rows_kernel: three launches
// of ~6 us each on the caller's stream.  `pos` is ascending (label compaction is row-major: compact_labels_kernel), so a row
// finds out whether it is a label row -- and which -- by bisection; no atomics, no ordering between workgroups.
// Reference op chain: the autograd of `x[non_pad_mask]` in NextItemPredictionTask.remove_pad_3d (prediction_task.py:472-479).
__global__ __launch_bounds__(256) void scatter_rows_dense_kernel(const float* __restrict__ src, const int* __restrict__ pos, int n,
                                                                  const float* __restrict__ scale, float* __restrict__ dx, long T,
                                                                  int D) {
    const int dq = D / 4;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * dq) return;
    const int t = (int)(i / dq), c = (int)(i % dq) * 4;
    int lo = 0, hi = n;                       // first index with pos[idx] >= t
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (pos[mid] < t) lo = mid + 1; else hi = mid;
    }
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lo < n && pos[lo] == t) {
        const float g = scale ? *scale : 1.f;
        const float4 v = *reinterpret_cast<const float4*>(src + (long)lo * D + c);
        o = make_float4(g * v.x, g * v.y, g * v.z, g * v.w);
    }
    *reinterpret_cast<float4*>(dx + (long)t * D + c) = o;
}
extern "C" int t4r_scatter_rows_dense(void* stream, const float* src, const int* pos, int n, const float* scale, float* dx, long T,
                                      int D) {
    if (T == 0) return 0;
    T4R_CHECK_ARG(D % 4 == 0 && dx && (n == 0 || (src && pos)), "scatter_rows_dense: D must be a multiple of 4, pointers non-null");
    const long t = T * (D / 4);
    hipLaunchKernelGGL(scatter_rows_dense_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, pos, n,
                       scale, dx, T, D);
    T4R_LAUNCH_CHECK();
    return 0;
}

// inference: position of the hidden state to score (prediction_task.py:453-461)
//   MLM: count(non-pad)   (index into the L+1 grid) ; otherwise count(non-pad) - 1 (wraps to L-1... torch -1)
__global__ __launch_bounds__(256) void last_positions_kernel(const long* __restrict__ ids, int B, int L,
                                                              int Lgrid, int is_mlm, long padding_idx,
                                                              int* __restrict__ pos) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    int n = 0;
    for (int l0 = 0; l0 < L; l0 += 64) {
        const int l = l0 + lane;
        n += __popcll(__ballot(l < L && ids[(long)b * L + l] != padding_idx));
    }
    int idx = is_mlm ? n : n - 1;
    if (idx < 0) idx += Lgrid;
    if (lane == 0) pos[b] = b * Lgrid + idx;
}
extern "C" int t4r_last_positions(void* stream, const long* item_ids, int B, int L, int Lgrid,
                                  int is_mlm, long padding_idx, int* pos) {
    if (B == 0) return 0;
    hipLaunchKernelGGL(last_positions_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       item_ids, B, L, Lgrid, is_mlm, padding_idx, pos);
    T4R_LAUNCH_CHECK();
    return 0;
}

// number of non-padding positions per session (+ extra), int32 [B]: the key_len of the opt-in attention
// padding mask (sessions are right-padded: utils/padding.py:48-68, so the valid keys are a prefix)
__global__ __launch_bounds__(256) void session_lengths_kernel(const long* __restrict__ ids, int B, int L,
                                                               int padding_idx, int extra, int* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int n = 0;
    for (int l = 0; l < L; ++l) n += ids[(long)b * L + l] != padding_idx ? 1 : 0;
    out[b] = n + extra;
}
extern "C" int t4r_session_lengths(void* stream, const long* item_ids, int B, int L, int padding_idx, int extra,
                                   int* out) {
    if (B <= 0) return 0;
    T4R_CHECK_ARG(item_ids && out && L >= 1, "session_lengths: bad arguments");
    hipLaunchKernelGGL(session_lengths_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, item_ids, B, L,
                       padding_idx, extra, out);
    T4R_LAUNCH_CHECK();
    return 0;
}
