/* t4r_hip.h -- C ABI of libt4r_hip.so: the MI355X (gfx950) implementation of the
 * Transformers4Rec session-sequence hot path
 *     tr.TabularSequenceFeatures -> tr.TransformerBlock (XLNet) -> tr.NextItemPredictionTask
 *
 * The reference (NVIDIA-Merlin/Transformers4Rec) is pure Python over ATen / HuggingFace: it has
 * NO native FFI for this path.  Each entry point below therefore replaces an ATen/HF *op chain*
 * of the reference; the chain it replaces is cited as  <reference file>:<lines>  (paths relative
 * to the reference repository root; "HF" = transformers/models/xlnet/modeling_xlnet.py, the
 * third-party dependency pinned by requirements/base_external.txt:1).  The Python binding a
 * maintainer adds is a ctypes stub: see INTEGRATION.md and transformers4rec_amd/_lib.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless marked "host"; tensors are dense row-major;
 *     float = fp32, long = int64 (torch.int64), unsigned char = torch.bool / uint8, int = int32
 *   - `stream` is a hipStream_t (0 = null stream); all work is enqueued, nothing synchronises
 *   - return 0 on success; -1 on an argument error, a hipError_t value on a launch error;
 *     t4r_last_error() gives the message (thread-local)
 *   - no global mutable state besides the (thread-local) error string, lazily-set kernel attributes, the
 *     precision / threshold switches set through t4r_set_* and the per-device side streams of the layer
 *     backward; in particular nothing is remembered from a forward call for its backward call: what a backward
 *     needs travels in caller-owned memory (workspaces, t4r_head_note);
 *     callable from any host thread with the right device current
 *   - "accumulated" outputs are read-modify-write (parameter gradients); everything else is
 *     overwritten
 */
#ifndef T4R_HIP_H
#define T4R_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

int t4r_abi_version(void);
const char* t4r_last_error(void);

/* ----------------------------------------------------------------------------------------
 * a1  ragged -> padded
 * replaces: transformers4rec/torch/utils/padding.py:48-68 (_pad_ragged_tensor: repeat_interleave,
 *           sparse_coo_tensor().to_dense(), F.pad) and :126-164 (pad_inputs length rule)
 * out[r, c] = values[offsets[r] + c] if c < min(len_r, L) else 0.   elem_size 4 (fp32) | 8 (int64)
 * t4r_ragged_max_len writes max_r(len_r) to *out_max (device int) for the
 * min(max_sequence_length, batch max) rule. */
int t4r_ragged_max_len(void* stream, const long* offsets, int rows, int* out_max);
int t4r_ragged_to_padded(void* stream, const void* values, const long* offsets, void* out, int rows,
                         int L, int elem_size);
/* Batch assembly of the device-resident feed (replaces the Merlin loader + pad_batch map,
 * utils/data_utils.py:216-494, utils/padding.py:72-122): the whole ragged column lives in HBM;
 * a batch is `rows` row ids (any order: shuffling), out[i, c] = values[offsets[row_ids[i]] + c],
 * right-zero-padded / truncated to L.  offsets == NULL with L = 1 gathers a scalar column. */
int t4r_ragged_gather_to_padded(void* stream, const void* values, const long* offsets,
                                const long* row_ids, void* out, int rows, int L, int elem_size);

/* ----------------------------------------------------------------------------------------
 * a2,a3,a5,a7,a8,a13  multi-feature embedding gather + aggregation (+ fused masking epilogue)
 * replaces: features/embedding.py:226-249 (per-feature nn.Embedding lookup),
 *           tabular/aggregation.py:35-47 (concat in sorted-name order: the host passes `col`),
 *           :140-157 (element-wise-sum), :162-193 (element-wise-sum-item-multi),
 *           masking.py:473-498 / :302-337 (apply_mask_to_inputs) when mask_mode != 0.
 * kind[f] 0: table lookup, input[f] = int64 ids [B*L_in], table[f] = [rows[f], dim[f]]
 *         1: dense rows, input[f] = fp32 [B*L_in, dim[f]] (soft embeddings, continuous pass-through)
 *         2: per-session table lookup (a3), input[f] = int64 ids [B], the row is broadcast over the
 *            sequence (features/embedding.py:229-240 2-D branch + tabular/base.py:53-63)
 *         3: per-session dense rows, input[f] = fp32 [B, dim[f]] (a context feature after a post
 *            transformation), broadcast over the sequence
 * agg 0 concat (col[f] = first output column) | 1 sum | 2 item * sum(others) (item_feat = index)
 * mask_mode 0 none | 1 MLM (out = mask ? memb : x) | 2 CLM train/eval (mask ? (l==L-1 ? 0 : x) : memb)
 *           | 3 CLM inference (mask ? x : memb).  L_out = L_in + 1 is the MLM-inference grid
 *           (position L duplicates L-1 before masking).
 * host arrays: kind, input, table, dim, col, rows (length n_feat <= 16).
 * *err_flag (device int, may be NULL) is set to 1 if an id is outside [0, rows). */
int t4r_seq_features_fwd(void* stream, int n_feat, const int* kind, const void* const* input,
                         const float* const* table, const int* dim, const int* col, const long* rows,
                         int agg, int item_feat, int B, int L_in, int L_out, int W, int mask_mode,
                         const unsigned char* mask, const float* masked_emb, float* out,
                         int* err_flag);
/* backward of one table lookup: d_table[id, :] += dout[tok, col:col+dim] for id != padding_idx
 * (nn.Embedding(padding_idx=0), features/sequence.py:75-81).  d_table accumulated.
 * ids_div = 1 (ids [B*L]) or L (per-session ids [B]: id index = tok / L). */
int t4r_embedding_bwd(void* stream, const float* dout, const long* ids, float* d_table, long ntok,
                      int W, int col, int dim, long rows, int padding_idx, int ids_div);
/* Deterministic form of the same backward (and the local "apply" of the row-sparse data-parallel
 * exchange): sort the lookups by row id once (stable radix sort; depends on the ids only, so it can run in
 * the forward pass), then a segmented sum in ascending lookup order -- one owner wave per table row, no
 * atomics.  Replaces ATen embedding_dense_backward behind features/embedding.py:226-249.
 *   t4r_sort_ids: keys_sorted[n] int32 ascending (padding_idx / out-of-range ids -> `rows`, last),
 *                 perm[n] int32 = index of the lookup; ws of t4r_sort_ids_ws_bytes(n) bytes.
 *   t4r_embedding_bwd_sorted: d_table[key, :] += sum_{lookups p with that key, ascending p}
 *                 sum_{l < ids_div} dout[(p*ids_div + l)*W + col : +dim]; ws of
 *                 t4r_embedding_bwd_sorted_ws_floats(n, dim) floats. */
long t4r_sort_ids_ws_bytes(long n);
int t4r_sort_ids(void* stream, const long* ids, long n, long rows, int padding_idx, int* keys_sorted,
                 int* perm, void* ws, long ws_bytes);
/* The same sort for the F tables of a multi-feature input block in ONE call (BASELINE configs[2]: item id + three more
 * categoricals = 4 x ~10 small launches on the caller's stream otherwise): ids = HOST array of F device pointers (n lookups
 * each), rows / padding_idx = host arrays [F]; feature f's result is the slice [f n, (f + 1) n) of keys_sorted / perm and
 * equals what t4r_sort_ids gives for that feature alone.  F <= 16; ws of t4r_sort_ids_multi_ws_bytes(n, F) bytes.  The
 * reference runs one embedding backward per table (features/embedding.py:226-249). */
long t4r_sort_ids_multi_ws_bytes(long n, int F);
int t4r_sort_ids_multi(void* stream, const long* const* ids, int F, long n, const long* rows, const int* padding_idx,
                       int* keys_sorted, int* perm, void* ws, long ws_bytes);
long t4r_embedding_bwd_sorted_ws_floats(long n, int dim);
int t4r_embedding_bwd_sorted(void* stream, const float* dout, const int* keys_sorted, const int* perm,
                             float* d_table, long n, int W, int col, int dim, long rows, int ids_div,
                             float* ws);
/* a3  EmbeddingBag branch: one bag of ids per row -> one combined row.
 * replaces: features/embedding.py:226-240 (EmbeddingFeatures.forward: 2-D ids, (values, offsets) tuples, 1-D ids)
 *           with :86-93 / :260-273 (EmbeddingBagWrapper(mode = TableConfig.combiner)) and :416-460 (the combiner
 *           option set "mean" | "sum" | "sqrtn"), i.e. torch.nn.EmbeddingBag WITHOUT a padding index: id 0 is an
 *           ordinary row, an empty bag gives a zero row.
 * matrix form: offsets == NULL, values [n_bags * fixed_k] (fixed_k = 1 for 1-D ids); ragged form: values [n_values],
 * offsets [n_bags], bag b = values[offsets[b] : offsets[b+1]) and the last bag runs to n_values.
 * combiner 0 sum | 1 mean (/ n_b) | 2 sqrtn (/ sqrt(n_b)).  out[b, col : col + dim] (row pitch ld_out floats).
 * *err (device int, may be NULL) is set to 1 on an id outside [0, rows).
 * t4r_embedding_bag_bwd_rows: the transpose as rows -- rows_out[i, :] = scale(bag of i) * dout[bag of i, col : col + dim]
 * for every member lookup i; the caller sums them into the table with t4r_sort_ids(values, padding_idx = -1) +
 * t4r_embedding_bwd_sorted (deterministic) or hands them to the row-sparse data-parallel exchange. */
int t4r_embedding_bag_fwd(void* stream, const float* table, long rows, int dim, const long* values,
                          const long* offsets, long n_bags, long n_values, int fixed_k, int combiner, float* out,
                          long ld_out, int col, int* err);
int t4r_embedding_bag_bwd_rows(void* stream, const float* dout, long ld, int col, int dim, const long* offsets,
                               long n_bags, long n_values, int fixed_k, int combiner, float* rows_out);
/* masking as its own pass (after the projection MLP), in place on x [B*L, H]; and its backward:
 * d_memb[H] += sum of dy over replaced tokens (accumulated), dy zeroed there (in place). */
int t4r_apply_mask_fwd(void* stream, float* x, const unsigned char* mask, const float* masked_emb,
                       int B, int L, int H, int mode);
long t4r_apply_mask_bwd_ws_floats(int B, int L, int H);
int t4r_apply_mask_bwd(void* stream, float* dy, const unsigned char* mask, float* d_masked_emb, int B,
                       int L, int H, int mode, float* ws /* t4r_apply_mask_bwd_ws_floats floats: fixed-order two-stage sum */);
/* the same out of place (dx = masked gradient, dy untouched; dy != dx): no clone of the incoming gradient in the caller */
int t4r_apply_mask_bwd_to(void* stream, const float* dy, float* dx, const unsigned char* mask, float* d_masked_emb, int B,
                          int L, int H, int mode, float* ws);
int t4r_mul(void* stream, const float* a, const float* b, float* out, long n);

/* a4  SoftEmbedding (+ per-feature LayerNorm)
 * replaces: features/embedding.py:551-556 (Linear(1,K) -> softmax -> weighted sum of E[K,D]) and
 *           :306-309 / tabular/transformations.py:128-132 (LayerNorm, eps 1e-5).  ln_w NULL = no LN.
 * K <= 32, D <= 32.  Backward accumulates all parameter gradients. */
int t4r_soft_embedding_fwd(void* stream, const float* x, const float* proj_w, const float* proj_b,
                           const float* table, const float* ln_w, const float* ln_b, float* out,
                           long ntok, int K, int D, float eps);
int t4r_soft_embedding_bwd(void* stream, const float* dout, const float* x, const float* proj_w,
                           const float* proj_b, const float* table, const float* ln_w,
                           float* d_proj_w, float* d_proj_b, float* d_table, float* d_ln_w,
                           float* d_ln_b, long ntok, int W, int col, int K, int D, float eps,
                           float* ws /* t4r_soft_embedding_bwd_ws_floats floats: per-workgroup partial sums, added in a fixed order (no atomics) */);
long t4r_soft_embedding_bwd_ws_floats(long ntok, int K, int D);

/* ----------------------------------------------------------------------------------------
 * a11,a12  masking schema + labels (integer, bit-exact)
 * replaces: masking.py:376-470 (MLM), :274-300 (CLM), :182-213 (predict_all)
 * mode 0 MLM train | 1 MLM eval last item | 2 MLM eval all | 3 MLM inference ([B, L+1] outputs)
 *      4 CLM train/eval all | 5 CLM last item only | 6 CLM inference
 * MLM train draws: bern [B,L] uint8, j1 [B], j2 [B] = the recorded torch.bernoulli /
 * torch.multinomial results (parity replay); pass NULL for device Philox draws (seed, offset).
 * row_count[B] (may be NULL) receives the number of labels per session. */
int t4r_mask_targets(void* stream, const long* item_ids, int B, int L, int mode, long padding_idx,
                     const unsigned char* bern, const long* j1, const long* j2,
                     float mlm_probability, unsigned long long seed, unsigned long long offset,
                     unsigned char* mask_schema, long* masked_targets, int* row_count);
/* a17  ordered compaction of label positions
 * replaces: model/prediction_task.py:436-443 + remove_pad_3d :472-479 (masked_select, row-major)
 * row_offset[B], *n_labels, label_pos[>= n_labels] (token index b*L+l), labels_compact[>= n_labels] */
int t4r_compact_labels(void* stream, const long* masked_targets, const int* row_count, int B, int L,
                       long padding_idx, int* row_offset, int* n_labels, int* label_pos,
                       long* labels_compact);
int t4r_gather_rows(void* stream, const float* x, const int* pos, float* out, int n, int D);
int t4r_scatter_rows_add(void* stream, const float* dout, const int* pos, float* dx, int n, int D);
/* dx [T, D] = 0 except dx[pos[r], :] = (*scale or 1) * src[r, :], r < n; pos ascending.  One launch for the backward of the
 * label-row selection x[non_pad_mask] (prediction_task.py:472-479): zero fill + upstream-gradient scaling + scatter. */
int t4r_scatter_rows_dense(void* stream, const float* src, const int* pos, int n, const float* scale, float* dx, long T, int D);
/* a21  inference row selection (prediction_task.py:453-461): pos[b] = b*Lgrid + last_item(b) */
int t4r_last_positions(void* stream, const long* item_ids, int B, int L, int Lgrid, int is_mlm,
                       long padding_idx, int* pos);

/* ----------------------------------------------------------------------------------------
 * dense contractions on the fp32 matrix cores (v_mfma_f32_32x32x2_f32)
 * replaces: every Linear / einsum / mm of the path: block/mlp.py:133-135, HF :253-259,:145,:297-305,
 *           model/prediction_task.py:664 (+ torch.div by temperature via alpha) and their autograd.
 * C[M,N] = alpha * op(A) * op(B) (+ epilogue).  transA=0: A is [M,lda] ; 1: A is [K,lda].
 * transB=0: B is [K,ldb] ; 1: B is [N,ldb].  epilogue 0 none | 1 +bias | 2 gelu(+bias), pre-act
 * to aux | 3 relu(+bias) | 4 dropout(x + bias) + residual (residual read from aux).
 * splitk 1: plain; >1 or -1 (auto): fp32 atomics into C (C zeroed first unless accumulate).
 * accumulate: C += result.  batch: strided batches.  drop_p > 0: dropout on the output of
 * epilogue 2 / the pre-residual value of epilogue 4, mask index row*N + col. */
int t4r_gemm_f32(void* stream, int transA, int transB, int M, int N, int K, float alpha,
                 const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                 const float* bias, int epilogue, float* aux, long ldaux, int splitk, int accumulate,
                 int batch, long strideA, long strideB, long strideC, float drop_p,
                 unsigned long long seed, unsigned long long ctr_hi);
/* Deterministic split-K for a caller's weight gradients (autograd of torch.nn.Linear's weight, block/mlp.py:133-135): between
 * _begin and _end every ACCUMULATING split-K t4r_gemm_f32 (dense C, epilogue 0) issued by THIS thread stores its partial tiles
 * into ws (cap_floats floats; M * N * splits per launch, up to 20 launches) instead of adding them with atomics, and _flush adds
 * them to C in split order with one launch -- the sum no longer depends on the order the workgroups finish.  A launch that
 * does not fit the workspace falls back to atomics.  The XLNet layer backward uses the same mechanism internally. */
void t4r_gemm_splitk_sink_begin(float* ws, long cap_floats);
int t4r_gemm_splitk_sink_flush(void* stream);
void t4r_gemm_splitk_sink_end(void);
/* launches since the last _begin (this thread) that were eligible for the sink but found no room and fell back to fp32 atomics */
int t4r_gemm_splitk_sink_bypassed(void);

/* Precision of every dense contraction launched after the call (process-wide setting; T4R_GEMM_PREC sets
 * the default):
 *   0  fp32 operands on the fp32 matrix cores -- the reference's own arithmetic
 *   1  fp32-accurate products on the BF16 matrix cores: each operand is cut into three bf16 pieces (exact
 *      split) while staged, six v_mfma_f32_32x32x16_bf16 partial products per K = 16, fp32 accumulation
 *   2  mixed precision, bf16 operands (round to nearest even), fp32 accumulation, fp32 outputs
 *   3  mixed precision, fp16 operands -- the reference's AMP mode (torch/trainer.py:363-367: HF Trainer
 *      fp16=True -> torch.cuda.amp.autocast; model/prediction_task.py:430); master weights stay fp32,
 *      LayerNorm / softmax / cross-entropy / the attention core stay fp32
 *   4  auto (default): fp32 accuracy, form 1 on the shapes where it measured faster than form 0
 * Launches whose operands are not 16-byte loadable, and the exact-rank evaluation epilogue, stay on form 0. */
void t4r_set_precision(int mode);
int t4r_get_precision(void);

/* Head backward with CrossEntropyLoss' backward fused into the A operand (the [N, V] gradient is
 * never written): dlogits = (*grad_out / n_rows) * (softmax(logits) - target), from logits/lse/labels.
 *   transA = 0: C[n_rows, N] (+)= alpha * dlogits @ B[V, N]       (d X)
 *   transA = 1: C[V, N]     (+)= alpha * dlogits^T @ B[n_rows, N] (d W)
 * replaces: the autograd of model/prediction_task.py:446 (CrossEntropyLoss) and :664 (X @ W^T). */
int t4r_gemm_softmax_grad_f32(void* stream, int transA, int n_rows, int V, int N, float alpha,
                              const float* logits, long ld_logits, const float* lse, const long* labels,
                              const float* grad_out, float label_smoothing, const float* B, long ldb,
                              float* C, long ldc, int splitk, int accumulate);

/* Row threshold of the token-stationary body GEMM (csrc/tok_gemm.hip: t4r_gemm_f32 takes it, in the fp32-accurate
 * precision modes, for op(A) = A with at least this many rows, N % 32 == 0, K in {32, 64, 96, 128, 256, 384, 512}).
 * Default 32 768 (T4R_TOK_GEMM_MIN_M); a negative value restores the default. */
void t4r_set_tok_gemm_min_rows(int rows);
int t4r_get_tok_gemm_min_rows(void);

/* Materialised tied full-softmax head for d_model 32 / 64 / 96 / 128, fp32-accurate on the BF16 matrix cores
 * (csrc/head_split.hip): the same three contractions as t4r_gemm_f32 (logits) and t4r_gemm_softmax_grad_f32
 * (d X, d W) of model/prediction_task.py:664 and its autograd through CrossEntropyLoss (:446), with the operand
 * cutting hoisted out of the inner loops.  ws: t4r_head_split_ws_bytes(N, V, D) bytes, 16-byte aligned, owned by the
 * caller from _prepare (forward) until the last backward product; _dx and _dw use disjoint parts of it and may run
 * on different streams.  logits [N, ld] holds the columns [yoff, yoff + Vc) of the [N, V] problem (Vc = V, yoff = 0:
 * all of them); W passed to _dx points at row yoff.  No atomics: results are bit-reproducible.
 * note (host, caller-owned, may be NULL): what the forward product tells its backward products -- which table slice the
 * max |W| word in ws describes, and whether ws holds the per-item column maxima of exactly these logits (then d W runs on
 * the two-way fp16 split with per-item scales, else on the three bf16 planes).  Zero it before the forward product and
 * pass the same struct to _dx / _dw of that forward; one note per forward in flight, not shared between concurrent calls.
 * t4r_head_note_dw_form(note): the form the last _dw with this note ran in -- 0 none yet, 1 three bf16 planes, 2 two-way
 * fp16 split with per-item scales. */
typedef struct t4r_head_note { unsigned long long w[8]; } t4r_head_note;
int t4r_head_note_dw_form(const void* note);
int t4r_head_split_supported(int D);
/* matrix instructions per fp32-equivalent one in the forward logits / d X products of csrc/head_split.hip: 3 = two-way fp16
 * split with power-of-two tensor scales (default), 6 = three bf16 planes (T4R_HEAD_FWD_FP16X2=0); d W always 6 */
int t4r_head_split_fwd_products(void);
long t4r_head_split_ws_bytes(int N, int V, int D);
int t4r_head_split_prepare(void* stream, const float* X, long ldx, int N, int D, int V, void* ws);
int t4r_head_split_logits(void* stream, void* ws, const float* W, long ldw, float* C, long ldc, int N, int V,
                          int D, float alpha, void* note);
/* logits + mean cross-entropy (label smoothing as losses.py:4-20) in ONE pass over the vocabulary: the logits are
 * stored as by _logits, the softmax statistics are reduced inside the product's workgroups (one partial per row and
 * 128-column tile in ws) and merged by a small kernel -- replaces t4r_softmax_ce_fwd's second pass over [N, V].
 * loss_rows [N], lse [N]; loss_mean (device scalar) may be NULL; labels == NULL runs the product (and its per-tile
 * statistics) alone. */
int t4r_head_split_logits_ce(void* stream, void* ws, const float* W, long ldw, float* C, long ldc, const long* labels,
                             float* loss_rows, float* lse, float* loss_mean, int N, int V, int D, float alpha,
                             float label_smoothing, void* note);
int t4r_head_split_dw(void* stream, void* ws, const float* logits, long ld, const float* lse,
                      const long* labels, const float* grad_out, float label_smoothing, float* dW, long lddw,
                      int N, int Vc, int V, int yoff, int D, float alpha, int accumulate, void* note);
int t4r_head_split_dx(void* stream, void* ws, const float* logits, long ld, const float* lse, const long* labels,
                      const float* grad_out, float label_smoothing, const float* W, long ldw, float* dX, long lddx,
                      int N, int Vc, int V, int yoff, int D, float alpha, int accumulate, void* note);

/* The ONE-PASS forward of the same head (round 5; csrc/head_split.hip: head_fwd_dx_kernel): logits, loss rows, lse, the mean
 * loss AND dX [N, D] = d (mean loss) / d X for grad_out = 1 from one launch over (128-row tile x item range) workgroups --
 * the score tile comes off the matrix cores once, is stored, and its probabilities against a running row reference feed the
 * d X product from registers (flash-attention style; a small kernel merges the per-range statistics and partial sums).  The
 * logits are then read ONCE more (t4r_head_split_dw) instead of twice: 3.65 -> 2.6 GB of HBM traffic per step at BASELINE
 * configs[1].  Replaces transformers4rec/torch/model/prediction_task.py:648-671 (logits) + CrossEntropyLoss :446 and the d X
 * half of their autograd.  X: the rows t4r_head_split_prepare was given (same ws); wsum: column sums of W [D], required when
 * label_smoothing > 0, else NULL; the caller's backward is dX * grad_out and t4r_head_split_dw with the same ws / note.
 * labels == NULL: the dominant kernel alone on a workspace a full call has filled (bench.py's roofline timing).
 * t4r_head_split_fdx_supported: 1 when this form takes the width (two-way fp16 products on; T4R_HEAD_FDX=0 switches it off). */
int t4r_head_split_fdx_supported(int D);
/* The next t4r_head_split_logits_ce_dx of this thread ON TABLE W takes max |W| from part[0 .. n) -- the per-workgroup maxima
 * t4r_adam_step_amax left over exactly W's elements; the caller promises that nothing has written W since -- instead of a memset
 * + a pass over the table.  Consumed by that call. */
void t4r_head_split_w_amax_hint(const float* W, const float* part, int n);
int t4r_head_split_logits_ce_dx(void* stream, void* ws, const float* X, long ldx, const float* W, long ldw, float* C, long ldc,
                                const long* labels, float* loss_rows, float* lse, float* loss_mean, float* dX, long lddx,
                                const float* wsum, int N, int V, int D, float alpha, float label_smoothing, void* note);
/* The RECOMPUTING form of the same head (round 4; two-way fp16 products only: t4r_head_split_recompute_supported): nothing
 * of size [N, V] is written or read.  _ce: loss rows, lse (and the mean) from per-tile statistics, each row's label logit
 * captured inside the product; _dw_rc / _dx_rc: the two backward products with their score tiles recomputed on the matrix
 * cores in the orientation whose accumulator layout is the A operand of the product that follows (1.1 GB of logits written
 * once and read twice per step at BASELINE configs[1] become two more products).  Same workspace (t4r_head_split_prepare),
 * same note discipline (the note _ce fills is REQUIRED by _dw_rc / _dx_rc), whole vocabulary only (no chunks); predictions,
 * when wanted, are t4r_head_split_logits on the same workspace.  X in _dx_rc: the rows _prepare was given.
 * t4r_head_note_dw_form reports 3 after _dw_rc.  _prepare_rc (after _prepare, same X and workspace, before _ce) cuts the one
 * more image of X that _dw_rc reads; the materialised head does not need it. */
int t4r_head_split_recompute_supported(int D);
int t4r_head_split_prepare_rc(void* stream, const float* X, long ldx, int N, int D, int V, void* ws);
int t4r_head_split_ce(void* stream, void* ws, const float* W, long ldw, const long* labels, float* loss_rows, float* lse,
                      float* loss_mean, int N, int V, int D, float alpha, float label_smoothing, void* note);
int t4r_head_split_dw_rc(void* stream, void* ws, const float* W, long ldw, const float* lse, const long* labels,
                         const float* grad_out, float label_smoothing, float* dW, long lddw, int N, int V, int D,
                         float alpha, int accumulate, void* note);
int t4r_head_split_dx_rc(void* stream, void* ws, const float* X, long ldx, const float* W, long ldw, const float* lse,
                         const long* labels, const float* grad_out, float label_smoothing, float* dX, long lddx, int N,
                         int V, int D, float alpha, int accumulate, void* note);

/* Non-materialising head: output projection + softmax cross-entropy WITHOUT an [N, V] logits tensor
 * (the form that can run a 10 M-item vocabulary: 15 k x 10 M logits would be 600 GB).
 * replaces: model/prediction_task.py:664-669 (logits = X @ W^T, / T) + :446 (CrossEntropyLoss(), mean;
 * torch/losses.py:4-20 label smoothing) and their autograd, when `predictions` are not requested.
 * The vocabulary is streamed in chunks of chunk_cols columns (multiple of 4) through one
 * [N, chunk] buffer of t4r_linear_softmax_ce_chunk_floats(N, chunk_cols) floats (size it to stay in the
 * 256 MB Infinity Cache); fwd keeps online (max, sum-exp) statistics (stats: 4*N floats scratch), bwd
 * recomputes each chunk and forms the softmax gradient inside the A operand of the two contractions.
 *   fwd: loss_rows[N], lse[N], *loss_mean (may be NULL) for softmax(alpha * X[N,D] @ W[V,D]^T) vs labels
 *   bwd: dX[N,D] = alpha * dlogits @ W (overwritten); dW[V,D] += alpha * dlogits^T @ X (NULL: skipped),
 *        dlogits = (*grad_out / N) * (softmax - (1-eps) onehot - eps/V); grad_out NULL = 1.
 * Every label must be in [0, V). */
long t4r_linear_softmax_ce_chunk_floats(int N, int chunk_cols);
int t4r_linear_softmax_ce_fwd(void* stream, const float* X, long ldx, const float* W, long ldw,
                              const long* labels, int N, int V, int D, float alpha, float label_smoothing,
                              int chunk_cols, float* chunk_buf, float* stats, float* loss_rows, float* lse,
                              float* loss_mean);
int t4r_linear_softmax_ce_bwd(void* stream, const float* X, long ldx, const float* W, long ldw,
                              const long* labels, const float* lse, const float* grad_out, int N, int V, int D,
                              float alpha, float label_smoothing, int chunk_cols, float* chunk_buf, float* dX,
                              long lddx, float* dW, long lddw);

/* residual + LayerNorm:  y = LN(a + b) (b may be NULL).
 * replaces: HF :142-152, :297-305 (post-LN), tabular/transformations.py:128-132.
 * backward recomputes x = a + b; dgamma/dbeta accumulated; dx overwritten (or += if accumulate_dx). */
/* Dropout (torch.nn.Dropout semantics) is fused where the reference applies it.  A keep decision is
 * a pure function of (seed, ctr_hi, element index): Philox4x32-10(key = seed, counter =
 * (index >> 2, ctr_hi)), component index & 3; kept values are scaled by 1/(1-p).  Nothing is stored:
 * backward kernels recompute the mask.  ctr_hi = t4r_dropout_ctr_hi(step offset, layer, site).
 * t4r_dropout: out = x[i % n_src] * mask(i)/(1-p) for i < n (n_src < n broadcasts x, e.g. pos_emb over
 * the batch); mask_out (uint8) optionally exports the keep mask; out may be NULL. */
unsigned long long t4r_dropout_ctr_hi(unsigned long long offset, int layer, int site);
int t4r_dropout(void* stream, const float* x, float* out, unsigned char* mask_out, long n, long n_src,
                float p, unsigned long long seed, unsigned long long ctr_hi);
/* y = LN(dropout(a) + b); mask index row*D + col.  drop_p = 0 disables. */
int t4r_add_layernorm_fwd(void* stream, const float* a, const float* b, const float* gamma,
                          const float* beta, float* y, float* mean, float* rstd, int rows, int D,
                          float eps, float drop_p, unsigned long long seed, unsigned long long ctr_hi);
/* dx = d loss / d b ; dxa = d loss / d a (= dx * mask/(1-p); may be NULL when drop_p == 0) */
int t4r_add_layernorm_bwd(void* stream, const float* a, const float* b, const float* gamma,
                          const float* mean, const float* rstd, const float* dy, float* dx, float* dxa,
                          float* dgamma, float* dbeta, float* ws, int rows, int D, int accumulate_dx,
                          float drop_p, unsigned long long seed, unsigned long long ctr_hi);
/* Batch-reduced gradients are summed in two deterministic stages through a caller workspace of
 * t4r_colreduce_ws_floats(rows, ncols) floats (ncols = 2*D for LayerNorm backward, N otherwise). */
long t4r_colreduce_ws_floats(long rows, int ncols);
/* activation backward + bias gradient: mode 0 GELU(erf) on saved pre-activation, 1 ReLU on saved
 * output; dbias (accumulated) may be NULL (then ws may be NULL).  N % 4 == 0. */
int t4r_act_bwd_bias(void* stream, const float* dact, const float* pre, float* dpre, float* dbias,
                     float* ws, long rows, int N, int mode, float drop_p, unsigned long long seed,
                     unsigned long long ctr_hi);
int t4r_colsum(void* stream, const float* x, float* out, float* ws, long rows, int N, long ld);

/* ----------------------------------------------------------------------------------------
 * a15  XLNet relative attention core and the whole layer
 * replaces: HF XLNetRelativeAttention.rel_attn_core :95-140 (+ rel_shift_bnij :81-93),
 *           XLNetLayer.forward :308-353 as configured by config/transformer.py:432-482.
 * q,k,v,out [B*L, n_head*d_head]; k_r [2L, D] = pos_emb @ r (kr_per_batch: [B,2L,D], one set per
 * session, used when pos_emb dropout is on); lse [B,n,L].  Any L >= 1 and any d_head up to 256
 * (the reference takes any total_seq_length / d_model, config/transformer.py:432-482): one-wave kernels (MFMA for L <= 32,
 * d_head 16 / 32) up to 64 positions with d_head 8 / 16 / 32, the general kernels of csrc/xlnet_attn_long.hip beyond (they
 * need `out` in the backward).  drop_p: attention-probability dropout (HF :132), mask index ((b*n+h)*L+i)*L+j.
 * backward: d_r_w_bias / d_r_r_bias accumulated, the rest overwritten (dk_r has k_r's shape).
 * key_len (device int32 [B], may be NULL = the reference's behaviour: NO padding mask): opt-in
 * padding mask -- keys at positions >= key_len[b] get the score -1e30 except on the diagonal, as HF XLNet does
 * when it is given an attention_mask (modeling_xlnet.py: attn_score - 1e30 * attn_mask, non_tgt_mask keeps
 * i == j).  t4r_session_lengths produces key_len from the item ids. */
int t4r_session_lengths(void* stream, const long* item_ids, int B, int L, int padding_idx, int extra, int* out);
int t4r_xlnet_attn_fwd(void* stream, const float* q, const float* k, const float* v, const float* k_r,
                       const float* r_w_bias, const float* r_r_bias, float* out, float* lse, int B,
                       int L, int n_head, int d_head, int kr_per_batch, float drop_p,
                       unsigned long long seed, unsigned long long ctr_hi, const int* key_len);
long t4r_xlnet_attn_bwd_ws_floats(int B, int L, int D, int n_head);
int t4r_xlnet_attn_bwd(void* stream, const float* q, const float* k, const float* v, const float* k_r,
                       const float* r_w_bias, const float* r_r_bias, const float* out,
                       const float* lse, const float* dout, float* dq, float* dk, float* dv,
                       float* dk_r, float* d_r_w_bias, float* d_r_r_bias, float* workspace, int B,
                       int L, int n_head, int d_head, int kr_per_batch, float drop_p,
                       unsigned long long seed, unsigned long long ctr_hi, const int* key_len);
/* a16  scaled-dot-product attention core of the GPT-2 (causal) and BERT blocks
 * replaces: HF gpt2/modeling_gpt2.py eager_attention_forward :54-72 ; HF bert BertSelfAttention.
 * q,k,v rows of `ld` floats (3*D for GPT-2's fused c_attn output), head h at columns h*d_head..;
 * out/dout rows of ld_out; lse [B,n,L]; any L >= 1 and any d_head up to 256 (LDS / MFMA kernels up to 128
 * positions with d_head 16|32|64, the general kernels of csrc/xlnet_attn_long.hip beyond).  key_len NULL = no padding mask (the
 * reference's behaviour); key_len int32 [B] (opt-in): keys >= key_len[b] are masked for every query, as HF does
 * with an attention_mask (finfo.min added to the scores).  drop_p: attention-probability dropout, mask index
 * ((b*n+h)*L+i)*L+j. */
int t4r_mha_fwd(void* stream, const float* q, const float* k, const float* v, long ld, float* out,
                long ld_out, float* lse, int B, int L, int n_head, int d_head, int causal, float drop_p,
                unsigned long long seed, unsigned long long ctr_hi, const int* key_len);
int t4r_mha_bwd(void* stream, const float* q, const float* k, const float* v, long ld, const float* out,
                const float* dout, long ld_out, const float* lse, float* dq, float* dk, float* dv, long ld_d,
                int B, int L, int n_head, int d_head, int causal, float drop_p, unsigned long long seed,
                unsigned long long ctr_hi, const int* key_len);
/* learned position (+ token-type row 0) embeddings: out[t] = x[t] + pos[t % L] (+ token_type);
 * backward accumulates d_pos[l] += sum_b dy[b,l]  (HF gpt2 :576-577 wpe ; HF bert embeddings) */
int t4r_add_pos_fwd(void* stream, const float* x, const float* pos, const float* token_type, float* out,
                    int B, int L, int D);
int t4r_add_pos_bwd(void* stream, const float* dy, float* d_pos, int B, int L, int D);

/* Token-tile-stationary fused kernels of the XLNet layer (csrc/xlnet_fused.hip, xlnet_fused_attn.hip; d_model 32 | 64 |
 * 128, d_inner = 4 d_model).  One workgroup owns 16 R token rows and runs a whole sub-block on them; fp32-accurate
 * products on the bf16 matrix cores (exact three-way operand cuts, six partial products, fp32 accumulation).  The weights
 * are cut ONCE per layer call into bf16 planes (t4r_xlnet_layer_prepare; `planes`: t4r_xlnet_layer_planes_floats(D)
 * floats, shared by every entry point below and by forward and backward).  t4r_xlnet_layer_fwd / _bwd use them by default
 * (T4R_XLNET_FUSED=0: the GEMM / element-wise launch chain).
 *
 * prepare: params = host array of the layer's 15 device pointers (order of t4r_xlnet_layer_fwd); one launch.
 *          t4r_xlnet_ff_prepare: the feed-forward planes only (same buffer layout).
 * qkv_proj:  q, k, v = h @ W_{q,k,v}  (HF modeling_xlnet.py:251-258, three einsum('ibh,hnd->ibnd')): qkv [3][T][D].
 * kr_proj:   k_r = pos @ r (HF :266 with :1142-1143): pos [rows, D] = the positional encoding [2L, D] or its per-session
 *            dropped copy [B 2L, D].
 * oproj_ln:  h1 = LayerNorm(dropout(attn_vec @ o^T) + h)  (HF post_attention :142-152).  Training form saves ao [T, D]
 *            (projection before dropout), mean, rstd [T]; inference form: all three NULL, drop_p = 0.
 * ln1_bwd:   dy = d loss / d h1 -> dh (residual part of d loss / d h), dao (rows of the o weight gradient
 *            d o += dao^T @ attn_vec, issued by the caller), dav = d loss / d attn_vec, all [T, D] overwritten;
 *            d_gamma, d_beta ACCUMULATED; part: t4r_xlnet_ln1_bwd_part_floats(T, D) floats.
 * dh:        dh [T, D] += d q @ W_q^T + d k @ W_k^T + d v @ W_v^T   (dqkv [3][T][D]).
 * ff_fwd:    replaces HF :297-305 XLNetFeedForward.forward -- layer_1, gelu(erf), dropout, layer_2, dropout,
 *            layer_norm(output + inp).  h1 [T, D] -> hout [T, D].  Training form: ffpre, ffact [T, 4D], ffout [T, D], mean,
 *            rstd [T] are all given and saved for the backward, the Philox masks of the two dropout sites are keyed
 *            (seed, ctr_act / ctr_out), element index row * width + col as everywhere.  Inference form: all five NULL.
 * ff_bwd:    dy [T, D] = d loss / d hout -> dh1 [T, D] (overwritten), plus the rows the two weight gradients contract over,
 *            dffout [T, D] and dpre [T, 4D] (overwritten; d W2 += dffout^T @ ffact, d W1 += dpre^T @ h1 are issued by the
 *            caller); d_gamma, d_beta, d_b2 [D], d_b1 [4D] are ACCUMULATED (per-workgroup partial sums + a reduction, no
 *            atomics).  part: t4r_xlnet_ff_bwd_part_floats(T, D) floats of scratch. */
int t4r_xlnet_fused_supported(int D);
/* matrix instructions per fp32-equivalent one in the fused layer kernels: 3 = two-way fp16 split with per-token /
 * per-matrix power-of-two scales (default), 6 = three bf16 planes (T4R_XLNET_FP16X2=0) */
int t4r_xlnet_fused_products(void);
long t4r_xlnet_layer_planes_floats(int D);
int t4r_xlnet_layer_prepare(void* stream, const float* const* params, int D, float* planes);
long t4r_xlnet_ff_planes_floats(int D);
int t4r_xlnet_ff_prepare(void* stream, const float* W1, const float* b1, const float* W2, int D, float* planes);
int t4r_xlnet_qkv_proj(void* stream, const float* h, const float* planes, float* qkv, long T, int D);
int t4r_xlnet_kr_proj(void* stream, const float* pos, const float* planes, float* kr, long rows, int D);
int t4r_xlnet_oproj_ln(void* stream, const float* av, const float* h, const float* planes, const float* gamma,
                       const float* beta, float* ao, float* mean, float* rstd, float* h1, long T, int D, float eps,
                       float drop_p, unsigned long long seed, unsigned long long ctr_hi);
long t4r_xlnet_ln1_bwd_part_floats(long T, int D);
int t4r_xlnet_ln1_bwd(void* stream, const float* dy, const float* ao, const float* h, const float* mean, const float* rstd,
                      const float* gamma, const float* planes, float* dh, float* dao, float* dav, float* d_gamma,
                      float* d_beta, float* part, long T, int D, float drop_p, unsigned long long seed,
                      unsigned long long ctr_hi);
int t4r_xlnet_dh(void* stream, const float* dqkv, const float* planes, float* dh, long T, int D);
/* CUs the token-tile kernels of the BACKWARD pass (t4r_xlnet_ff_bwd, _ln1_bwd, _dh; also inside t4r_xlnet_layer_bwd) may count
 * on; 0 (default) = the whole chip.  Their default grid is one 512-thread workgroup per CU, so a resident kernel that holds k
 * CUs -- the RCCL all-reduce of the table bucket that runs under the body's backward at N > 1 (the reference's
 * DDP does the same overlap, transformers4rec/torch/trainer.py:131-161) -- would send every such launch into a second round
 * of workgroups (measured 1.38x per step, tools/occupier_curve.py).  With a budget below 256 they take the tile that minimises
 * rounds x rows on that many CUs.  Process-wide, read at every launch.  Per-row results (d h, d attn_out, ...) are bit-identical
 * whatever the budget; the batch-reduced gradients (d gamma, d beta, d b1, d b2) are equal UP TO SUMMATION ORDER: the tile size
 * sets how the per-workgroup partial sums are grouped, so a budgeted backward rounds them differently from an unbudgeted one
 * (each is still bit-reproducible for its own budget).  t4r_device_cus(): compute units of the current device (the budget's
 * upper bound; replaces hipDeviceGetAttribute(hipDeviceAttributeMultiprocessorCount) for the caller). */
void t4r_xlnet_set_cu_budget(int cus);
int t4r_xlnet_get_cu_budget(void);
int t4r_device_cus(void);
/* 1 if the library was built with -DT4R_EXPERIMENTAL (A/B switches readable from the environment: csrc/t4r_common.h), else 0 */
int t4r_experimental_build(void);
/* The attention half of a layer as ONE kernel per direction (csrc/xlnet_attn_block.hip; round 4): exact fp32 matrix
 * instructions (v_mfma_f32_16x16x4_f32 = a k-ordered fmaf chain), one workgroup per 80 / L whole sessions, q | k | v, the
 * scores and the probabilities never leave the chip.  Shapes: L <= 32, d_head 16 | 32, d_model 32 | 64 | 128
 * (t4r_xlnet_attn_block_supported); t4r_xlnet_layer_fwd uses it by default there (T4R_XLNET_ATTN_BLOCK=0: the
 * qkv_proj -> attention core -> oproj_ln launches above).
 * fwd replaces HF modeling_xlnet.py XLNetRelativeAttention.forward :245-282 (g = None) + rel_attn_core :96-140 +
 *     post_attention :142-152:  h [T, D] -> h1 [T, D] = LayerNorm(dropout(attn_vec @ o^T) + h).
 *     planes: t4r_xlnet_layer_prepare's buffer (the fp32 transposes at its end); o: the o weight [D][n_head d_head] as
 *     stored; kr: k_r = pos_emb @ r, [2L][D] (kr_bstride 0) or per session [B][2L][D] (kr_bstride 2 L D).
 *     Saved for the backward: qkv [3][T][D], av = attn_vec [T][D], lse [B][n_head][L]; training (ao != NULL): ao [T][D]
 *     (o-projection before dropout), mean, rstd [T].  Dropout sites keyed (seed, ctr_prob) on the probabilities
 *     (element ((b n_head + head) L + i) L + j) and (seed, ctr_out) on the projection (element t D + feature), the same
 *     keys and element indices as t4r_xlnet_attn_fwd / t4r_xlnet_oproj_ln: the masks are identical.  key_len: optional
 *     int32 [B] (opt-in padding mask as t4r_xlnet_attn_fwd).
 * The backward of the attention half stays three launches (t4r_xlnet_ln1_bwd -> t4r_xlnet_attn_bwd -> t4r_xlnet_dh); a
 *     one-kernel backward was built, tested and measured slower (docs/DESIGN_rounds_1_to_4.md, round 4) -- tools/experimental/, not exported. */
int t4r_xlnet_attn_block_supported(int L, int D, int n_head);
int t4r_xlnet_attn_block_fwd(void* stream, const float* h, const float* planes, const float* o, const float* kr,
                             long kr_bstride, const float* r_w_bias, const float* r_r_bias, const float* gamma,
                             const float* beta, float* qkv, float* av, float* lse, float* ao, float* mean, float* rstd,
                             float* h1, int B, int L, int D, int n_head, float eps, float drop_p, unsigned long long seed,
                             unsigned long long ctr_prob, unsigned long long ctr_out, const int* key_len);
long t4r_xlnet_ff_bwd_part_floats(long T, int D);
int t4r_xlnet_ff_fwd(void* stream, const float* h1, const float* planes, const float* b1, const float* b2,
                     const float* gamma, const float* beta, float* ffpre, float* ffact, float* ffout, float* mean,
                     float* rstd, float* hout, int T, int D, float eps, float drop_p, unsigned long long seed,
                     unsigned long long ctr_act, unsigned long long ctr_out);
int t4r_xlnet_ff_bwd(void* stream, const float* dy, const float* ffout, const float* h1, const float* mean,
                     const float* rstd, const float* gamma, const float* ffpre, const float* planes, float* dh1,
                     float* dffout, float* dpre, float* d_gamma, float* d_beta, float* d_b2, float* d_b1, float* part,
                     int T, int D, float drop_p, unsigned long long seed, unsigned long long ctr_act,
                     unsigned long long ctr_out);

/* params / grads: host arrays of 15 device pointers in the order
 *   q, k, v, o, r [D,n,dh] ; r_w_bias, r_r_bias [n,dh] ; rel_attn.layer_norm.{weight,bias} ;
 *   ff.layer_1.{weight [4D,D], bias} ; ff.layer_2.{weight [D,4D], bias} ; ff.layer_norm.{weight,bias}
 * (state_dict names of SURVEY 8(b)).  pos_emb [2L, D] = HF relative_positional_encoding :940-976.
 * ws: t4r_xlnet_layer_ws_floats() floats saved by fwd for bwd; bws: scratch for bwd.
 * drop_p > 0 enables the reference's training-mode dropouts of the layer (HF :132,:147,:301,:303 and
 * the per-session pos_emb dropout :1143) with masks keyed by (seed, offset = step counter, layer_idx);
 * the workspace queries take dropout = (drop_p > 0).
 * layer_idx: bits 0-7 the layer (the only part the mask keys use); bit 8 (0x100), with drop_p > 0 and d_model 32 / 64 / 128:
 * this is the LAST layer of the stack and its feed-forward kernels also apply the MODEL's output dropout (HF :1177, key
 * (offset, 255, site 6)) -- to h_out in _fwd, to dh_out on load in _bwd -- instead of two element-wise launches over [T, D]
 * around the stack (round 6); bit 9 (0x200), with drop_p > 0, d_model 32 / 64 / 128 and t4r_xlnet_attn_block_supported(L, D,
 * n_head): this is the FIRST layer and `h` is the model's UNDROPPED input -- the input dropout (HF :1116, key (offset, 255,
 * site 0)) is applied on load by the attention-block kernel (the dropped rows are kept in `ws` for _bwd, which takes them from
 * there whatever `h` it is given) and _bwd masks the d h it returns. */
long t4r_xlnet_layer_ws_floats(int B, int L, int D, int n_head, int dropout);
long t4r_xlnet_layer_bwd_ws_floats(int B, int L, int D, int n_head, int dropout);
int t4r_xlnet_layer_fwd(void* stream, const float* h, const float* pos_emb, const float* const* params,
                        float* ws, float* h_out, int B, int L, int D, int n_head, float ln_eps,
                        float drop_p, unsigned long long seed, unsigned long long offset, int layer_idx,
                        const int* key_len, const float* pos_emb_b);
int t4r_xlnet_layer_bwd(void* stream, const float* h, const float* pos_emb, const float* const* params,
                        float* const* grads, const float* ws, float* bws, const float* dh_out,
                        float* dh_in, int B, int L, int D, int n_head, float ln_eps, float drop_p,
                        unsigned long long seed, unsigned long long offset, int layer_idx, const int* key_len,
                        const float* pos_emb_b);
/* Prologue of a whole XLNet layer stack (csrc/xlnet_fused_attn.hip, csrc/xlnet_layer.hip): the weight planes of ALL
 * layers in one launch and the positional keys k_r = pos @ r_l of all layers in one launch per four layers (HF
 * modeling_xlnet.py:266 k_head_r; the positional encoding and its dropout mask :1143 are shared by the layers), instead
 * of two launches inside every t4r_xlnet_layer_fwd.  params_all: n_layers x 15 device pointers (order of
 * t4r_xlnet_layer_fwd); planes[l] / kr[l]: inside layer l's workspace at t4r_xlnet_layer_ws_offsets' offsets; pos: the
 * [pos_rows, D] positional encoding the layers would project (pos_emb_b [B 2L, D] with dropout, pos_emb [2L, D] without).
 * After t4r_xlnet_stack_prepared(1) the layer forwards of this thread skip their own two launches (fused widths only;
 * pos_emb_b must be passed to them when dropout is on). */
int t4r_xlnet_layer_ws_offsets(int B, int L, int D, int n_head, int dropout, long* planes_off, long* kr_off);
int t4r_xlnet_stack_prepare(void* stream, const float* const* params_all, int n_layers, int D, float* const* planes,
                            const float* pos, long pos_rows, float* const* kr);
/* The next t4r_xlnet_stack_prepare of this thread makes the dropped positional rows of a TRAINING forward itself: it is then
 * given the plain [period, D] encoding as `pos` with pos_rows = B x period, masks row t = pos[t % period] with the pos_emb
 * dropout keyed (seed, ctr) (HF modeling_xlnet.py:1143) inside the projection kernel and writes the dropped rows to out
 * [pos_rows, D] -- the tensor the layers' backward takes as pos_emb_b.  Replaces a t4r_dropout launch in front of the call. */
void t4r_xlnet_stack_pos_dropout(float p, unsigned long long seed, unsigned long long ctr, long period, float* out);
void t4r_xlnet_stack_prepared(int on);

/* Deferred join of the layer backward's weight-gradient streams (csrc/xlnet_layer.hip).  After
 * t4r_xlnet_layer_bwd_defer(1) a t4r_xlnet_layer_bwd call on this thread returns without ordering `stream` after its
 * weight gradients (q, k, v, o, r, W1, W2 and the bias / LayerNorm sums may still be accumulating); the caller keeps
 * every buffer it passed alive and unwritten and calls t4r_xlnet_layer_bwd_join(stream) before the gradients are read
 * or those buffers are reused.  Default 0: every call joins before it returns (reference semantics: autograd returns
 * finished gradients, transformers4rec/torch/block/transformer.py:179-199 under torch.autograd). */
void t4r_xlnet_layer_bwd_defer(int on);
int t4r_xlnet_layer_bwd_join(void* stream);


/* ----------------------------------------------------------------------------------------
 * a18-a21  next-item head
 * replaces: model/prediction_task.py:347,446 (CrossEntropyLoss, mean), torch/losses.py:4-20 (label
 *           smoothing), :673-696 (sampled softmax logits), :466-470 (torch.topk).
 * logits [N, ld] (ld >= V, leading dimension); loss_rows/lse [N]; *loss_mean = mean(loss_rows).
 * bwd: dlogits = (*grad_out / N) * (softmax - target), pad columns V..ld-1 written 0. */
int t4r_softmax_ce_fwd(void* stream, const float* logits, const long* labels, float* loss_rows,
                       float* lse, float* loss_mean, int N, int V, long ld, float label_smoothing);
int t4r_softmax_ce_bwd(void* stream, const float* logits, const long* labels, const float* lse,
                       const float* grad_out, float* dlogits, int N, int V, long ld,
                       float label_smoothing);
/* ws: n_neg * D floats (the negatives' scores then are one X @ W_neg^T contraction on the matrix cores); NULL: row-wise */
int t4r_sampled_logits_fwd(void* stream, const float* x, const long* labels, const float* W,
                           const long* neg_samples, const float* sampling_dist, float* out, int N,
                           int D, int n_neg, float temperature, float* ws);
/* n draws (with replacement) from LogUniformSampler's distribution over [min_id, max_id) (prediction_task.py:766-786,
 * 843-848: torch.multinomial(self.dist, n_tries, replacement=True)) by the closed-form inverse CDF
 * id = min_id + floor(R^u) - 1, R = max_id - min_id + 1, u = Philox(seed, (i, ctr_hi)); out int64 [n] */
int t4r_log_uniform_sample(void* stream, long* out, int n, long min_id, long max_id, unsigned long long seed,
                           unsigned long long ctr_hi);
/* dlogits is modified in place (accidental-hit entries zeroed: they carry no gradient);
 * ws = 2 * n_neg * D floats of scratch (gathered negative rows of W and their gradient). */
int t4r_sampled_logits_bwd(void* stream, float* dlogits, const float* x, const long* labels,
                           const float* W, const long* neg_samples, float* dx, float* dW, float* ws, int N,
                           int D, int n_neg, float temperature);
/* row-sparse form of the same backward: the weight gradient is returned as rows instead of being added
 * into a dense [V, D] buffer -- rows_out[(N + n_neg), D] = gradient rows of the ids labels[0..N) ++
 * neg_samples[0..n_neg).  The caller sums them into the table (t4r_sort_ids + t4r_embedding_bwd_sorted,
 * deterministic) or exchanges them between data-parallel ranks (all_gather(ids, rows)). */
int t4r_sampled_logits_bwd_rows(void* stream, float* dlogits, const float* x, const long* labels,
                                const float* W, const long* neg_samples, float* dx, float* rows_out,
                                float* ws, int N, int D, int n_neg, float temperature);
/* 1 <= k <= min(256, V); values descending, ties to the lower index (as torch.topk on distinct values / a stable sort) */
int t4r_topk(void* stream, const float* scores, int N, int V, long ld, int k, float* out_val,
             long* out_idx);
/* Fused eval head (replaces logits materialisation + torch.topk + the [N, V] one-hot of
 * ranking_metric.py:52-59 for the metric computation): rank[row] = number of items that beat the
 * row's target under "ties go to the lower index", from alpha * X[n_rows, D] @ W[V, D]^T computed
 * tile by tile and never stored.  target_score[row] = that row's score of labels[row], produced by
 * the same GEMM kernel (t4r_gemm_f32 on the gathered label rows, diagonal).  Recall@k = rank < k,
 * NDCG@k = rank < k ? 1 / log2(rank + 2) : 0. */
int t4r_rank_of_target_f32(void* stream, int n_rows, int V, int D, float alpha, const float* X, long ldx,
                           const float* W, long ldw, const float* target_score, const long* labels,
                           int* rank);

/* ----------------------------------------------------------------------------------------
 * train-time input regularisers (pre / post transformations of the input block)
 *
 * t4r_swap_noise: tr.StochasticSwapNoise.augment for one feature
 * (transformers4rec/torch/tabular/transformations.py:55-93).  x/out hold n elements of elem_bytes
 * (8: int64 ids, 4: fp32 continuous values).  item_ids[i * mask_stride] != pad_token is the padding
 * mask of element i (config/schema.py:59-66; mask_stride = L for a per-session [B] feature, which
 * uses mask[:, 0]; 1 otherwise).
 * Element i is replaced when it is non-pad and its Bernoulli(p) trial succeeds; the k-th replaced
 * element (row-major) receives masked[perm[k]], masked = the non-pad values in row-major order.
 * bern (uint8 [n]) and perm (int64 [#non-pad]) inject the draws (parity tests replay the
 * reference's torch.bernoulli / torch.randperm); null => device draws from Philox(seed, ctr_hi)
 * (uniforms for the trial, a radix sort of 64-bit keys for the permutation).
 * ws: t4r_swap_noise_ws_bytes(n) bytes of device scratch.
 *
 * TabularDropout (transformations.py:145-160) is t4r_dropout; the per-feature TabularLayerNorm
 * (transformations.py:96-142, eps 1e-5) is t4r_add_layernorm_fwd/bwd with b = NULL on the
 * feature's [tokens, dim] rows.  t4r_copy_cols moves one feature's column block between a
 * concatenated row buffer (`wide`, row pitch ldw) and a dense [rows, dim] buffer
 * (dir 0: wide -> narrow, 1: narrow -> wide). */
long t4r_swap_noise_ws_bytes(long n);
int t4r_swap_noise(void* stream, const void* x, void* out, int elem_bytes, long n,
                   const long* item_ids, long pad_token, long mask_stride, float p,
                   const unsigned char* bern, const long* perm, unsigned long long seed, unsigned long long ctr_hi, void* ws,
                   long ws_bytes);
int t4r_copy_cols(void* stream, float* wide, long ldw, int col, float* narrow, int dim, long rows,
                  int dir);
/* out[b, 0:dim] = sum_l wide[b*L + l, col:col+dim]: gradient of broadcasting a per-session feature
 * over the sequence (tabular/base.py:53-63) when a post transformation sits in between. */
int t4r_seq_sum_cols(void* stream, const float* wide, long ldw, int col, float* out, int dim, long B,
                     int L);

/* ----------------------------------------------------------------------------------------
 * optimizer: fused Adam over a flat parameter buffer (torch.optim.Adam semantics, the optimizer
 * of the reference's Model.fit, torch/model/base.py:669-718).  grad is multiplied by grad_scale
 * first (1/world_size after the RCCL sum); zero_grad clears grad in the same pass. */
int t4r_adam_step(void* stream, float* param, float* grad, float* exp_avg, float* exp_avg_sq, long n,
                  int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                  float grad_scale, int zero_grad);
/* the same step that also leaves amax_part[b] = max |param[i]| AFTER the update over i in [amax_lo, amax_hi) seen by workgroup b;
 * returns the number of workgroups (<= 1024: the capacity amax_part must have) or < 0.  With t4r_head_split_w_amax_hint it
 * replaces the memset + 21 us pass over the tied item table in front of every step's head (round 6). */
int t4r_adam_step_amax(void* stream, float* param, float* grad, float* exp_avg, float* exp_avg_sq, long n, int step, float lr,
                       float beta1, float beta2, float eps, float weight_decay, float grad_scale, int zero_grad, long amax_lo,
                       long amax_hi, float* amax_part);

#ifdef __cplusplus
}
#endif
#endif /* T4R_HIP_H */
