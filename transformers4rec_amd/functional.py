"""One training step of the hot path composed of REGISTERED operators only (`torch.ops.t4r_hip.*`, torch_ops.py): every
parameter gradient flows through autograd as an operator output, so

  * `torch.autograd.grad` / `.backward()` populate `.grad` through AccumulateGrad -- autograd hooks fire, torch DDP can wrap
    the model (the module mirror's own backward writes into flat buffers and bypasses them: dropin.py's guard);
  * `make_fx` of the whole step (forward + backward) yields a graph of `t4r_hip::*` nodes that replays to the same numbers
    (the reference pins traced == eager for its modules: tests/unit/torch/model/test_model.py:58-91).

Scope: BASELINE configs[0] / [1] -- item-id sequence, XLNet body, MLM, tied full softmax (the path the benchmark runs).
The arithmetic is the module mirror's (same kernels through ops.py); the head takes the general contraction kernels
(gemm + softmax-CE + softmax-gradient operand), not head_split.hip's workspace form.

    params = functional.parameters_of(model)            # name -> nn.Parameter, shared with the model
    loss, n_labels = functional.mlm_step(params, cfg_of(model), ids, rng)
"""
from typing import Dict

import torch

from . import ops, torch_ops  # noqa: F401  (registers the library)
from .transformer import relative_positional_encoding

T4R = torch.ops.t4r_hip


def layer_params(model):
    return [[q for q in lay.ordered_params()] for lay in model.transformer_block.transformer.layer]


def config_of(model):
    t = model.transformer_block.transformer
    m = model.input_features.masking
    return dict(n_head=t.config.n_head, eps=float(t.config.layer_norm_eps), dropout=float(t.config.dropout),
                mlm_probability=float(m.mlm_probability), padding_idx=int(m.padding_idx),
                temperature=float(model.prediction_task.pre.module.softmax_temperature or 1.0),
                label_smoothing=float(getattr(model.prediction_task.loss, "label_smoothing", 0.0) or 0.0))


def mlm_step(table: torch.Tensor, masked_emb: torch.Tensor, layers, cfg: Dict, ids: torch.Tensor, mask_seed: int, mask_offset: int,
             drop_seed: int = 0, drop_offset: int = 0, training: bool = True, n_labels: int = None):
    """-> (loss, label rows).  table [V, D] (tied), masked_emb [D], layers: per layer the 15 tensors in
    ops.XLNET_PARAM_ORDER.  dropout > 0 is not wired here (the model-level input / final sites and the per-session
    positional dropout live in transformer.XLNetModel): pass a model built with dropout 0 or cfg['dropout'] = 0.
    n_labels: the number of label rows when the caller already knows it (a traced step: the count shapes the head's
    operands, so a trace is specialised to it; None reads it from the device, the step's one host read)."""
    if cfg.get("dropout", 0.0) > 0 and training:
        raise NotImplementedError("functional.mlm_step: dropout sites of XLNetModel are not part of the functional path yet")
    B, L = ids.shape
    D = table.shape[1]
    mask, labels, pos, lab, n = T4R.mlm_targets(ids, cfg["mlm_probability"], mask_seed, mask_offset, cfg["padding_idx"])
    x = T4R.seq_item_embedding(ids, table, mask, masked_emb, ops.MASK_MLM)
    h = x.view(B * L, D)
    pe = relative_positional_encoding(L, D).to(h.device).contiguous()
    for li, p in enumerate(layers):
        h, _ws = T4R.xlnet_layer_fwd(h, pe, list(p), B, L, cfg["n_head"], cfg["eps"], 0.0, drop_seed, drop_offset, li)
    N = int(n.item()) if n_labels is None else int(n_labels)      # the one host read of the step (the module mirror reads it the same way)
    rows = T4R.gather_label_rows(h, pos, N)
    loss, _logits, _lse = T4R.linear_softmax_ce(rows, table, lab[:N].contiguous(), 1.0 / cfg["temperature"], cfg["label_smoothing"])
    return loss, N


class FunctionalMLMModel(torch.nn.Module):
    """The model's OWN parameters behind a forward made of registered operators only: what torch DDP (or any tool that
    relies on autograd hooks) can wrap -- `DistributedDataParallel(FunctionalMLMModel(model))` reduces every gradient through
    its bucket hooks, where the module mirror's flat-buffer backward (and the drop-in built on it) must exchange gradients
    itself.  Parameters are shared with `model` (no copy), so optimizers / checkpoints of `model` keep working.
    forward(ids) -> {"loss", "n_labels"}; device draws of the MLM mask advance as in the module mirror."""

    def __init__(self, model):
        super().__init__()
        self.cfg = config_of(model)
        m = model.input_features.masking
        self.table = model.input_features.item_embedding_table.weight
        self.masked_emb = m.masked_item_embedding
        self.layers = torch.nn.ModuleList([torch.nn.ParameterList(p) for p in layer_params(model)])
        self._masking = [m]           # not a sub-module: only its seed / offset are read and advanced

    def forward(self, ids):
        m = self._masking[0]
        loss, n = mlm_step(self.table, self.masked_emb, [list(p) for p in self.layers], self.cfg, ids, m.seed, m._rng_offset,
                           training=self.training)
        m._rng_offset += ids.numel()
        return {"loss": loss, "n_labels": n}
