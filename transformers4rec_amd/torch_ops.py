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
reference pins for its own modules (tests/unit/torch/test_torchscript.py:26).  Training keeps its autograd.Functions:
their backward writes parameter gradients straight into flat `.grad` buffers, which a functional operator cannot.
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
                    eps: float, drop_p: float, seed: int, offset: int, layer_idx: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """training form: -> (out [B*L, D], workspace of saved activations for xlnet_layer_bwd)"""
    out, ws = ops.xlnet_layer_fwd(h.contiguous(), pos_emb, [p.detach().contiguous() for p in params], B, L, n_head, eps,
                                  drop_p=drop_p, seed=seed, offset=offset, layer_idx=layer_idx)
    return out, ws


@xlnet_layer_fwd.register_fake
def _(h, pos_emb, params, B, L, n_head, eps, drop_p, seed, offset, layer_idx):
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


OPERATORS = ("gemm", "item_scores", "topk", "rank_of_target", "embedding_gather", "embedding_bag", "ragged_to_padded",
             "xlnet_layer_infer", "xlnet_layer_fwd", "xlnet_layer_bwd")
