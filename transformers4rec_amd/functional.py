"""One training step of the hot path composed of REGISTERED operators only (`torch.ops.t4r_hip.*`, torch_ops.py): every
parameter gradient flows through autograd as an operator output, so

  * `torch.autograd.grad` / `.backward()` populate `.grad` through AccumulateGrad -- autograd hooks fire, torch DDP can wrap
    the model (the module mirror's own backward writes into flat buffers and bypasses them: dropin.py's guard);
  * `make_fx` of the whole step (forward + backward) yields a graph of `t4r_hip::*` nodes that replays to the same numbers
    (the reference pins traced == eager for its modules: tests/unit/torch/model/test_model.py:58-91).

Scope: BASELINE configs[0] / [1] -- item-id sequence, XLNet body, MLM, tied full softmax, dropout as XLNetConfig.build
sets it (0.3): the configuration the benchmark runs, every dropout site included (round 5).  The arithmetic is the module
mirror's (same kernels through ops.py, same Philox keys): at a head_split.hip shape the head is the one-pass operator
(`t4r_hip::next_item_head`: logits + CE + d X from one launch), elsewhere the general contraction kernels.

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


def _head_split_form(rows, table, N):
    """the shapes the module mirror sends to csrc/head_split.hip's one-pass head (prediction_task._head_split_ok)"""
    from .prediction_task import _head_split_ok

    V, D = table.shape
    return ops.head_split_fdx_supported(D) and _head_split_ok(rows, table, N, V)


def mlm_step(table: torch.Tensor, masked_emb: torch.Tensor, layers, cfg: Dict, ids: torch.Tensor, mask_seed: int, mask_offset: int,
             drop_seed: int = 0, drop_offset: int = 0, training: bool = True, n_labels: int = None):
    """-> (loss, label rows).  table [V, D] (tied), masked_emb [D], layers: per layer the 15 tensors in
    ops.XLNET_PARAM_ORDER.  cfg['dropout'] > 0 (training): every dropout site of HF XLNetModel as the module mirror places
    them (transformer.XLNetModel.forward) -- inputs_embeds, the per-session positional encodings (drawn once, shared by the
    layers), the in-layer sites, the output -- with Philox keys (drop_seed, drop_offset): the SAME masks as the module mirror
    draws for (seed, _drop_offset), so the two paths agree mask for mask.
    n_labels: the number of label rows when the caller already knows it (a traced step: the count shapes the head's
    operands, so a trace is specialised to it; None reads it from the device, the step's one host read).
    training=False: the evaluation protocol (last item of every session masked, masking.py:461-465), no dropout."""
    B, L = ids.shape
    D = table.shape[1]
    p = float(cfg.get("dropout", 0.0)) if training else 0.0
    if training:
        mask, labels, pos, lab, n = T4R.mlm_targets(ids, cfg["mlm_probability"], mask_seed, mask_offset, cfg["padding_idx"])
    else:
        mask, labels, counts = ops.mask_targets(ids.contiguous(), ops.MLM_EVAL_LAST, cfg["padding_idx"])
        n, pos, lab = ops.compact_labels(labels, counts, cfg["padding_idx"])
    x = T4R.seq_item_embedding(ids, table, mask, masked_emb, ops.MASK_MLM, int(cfg["padding_idx"]))
    pe = relative_positional_encoding(L, D).to(x.device).contiguous()
    pos_b = None
    if p > 0:
        x = T4R.dropout(x, p, drop_seed, ops.dropout_ctr_hi(drop_offset, 255, ops.SITE_INPUT))
        pos_b = T4R.pos_emb_dropout(pe, B, p, drop_seed, drop_offset)
    h = x.view(B * L, D)
    for li, prm in enumerate(layers):
        h, _ws = T4R.xlnet_layer_fwd(h, pe, list(prm), B, L, cfg["n_head"], cfg["eps"], p, drop_seed, drop_offset, li, pos_b)
    if p > 0:
        h = T4R.dropout(h, p, drop_seed, ops.dropout_ctr_hi(drop_offset, 255, ops.SITE_FINAL))
    N = int(n.item()) if n_labels is None else int(n_labels)      # the one host read of the step (the module mirror reads it the same way)
    rows = T4R.gather_label_rows(h, pos, N)
    y = lab[:N].contiguous()
    if _head_split_form(rows, table, N):
        loss = T4R.next_item_head(rows, table, y, 1.0 / cfg["temperature"], cfg["label_smoothing"])[0]
    else:
        loss = T4R.linear_softmax_ce(rows, table, y, 1.0 / cfg["temperature"], cfg["label_smoothing"])[0]
    return loss, N


class FunctionalMLMModel(torch.nn.Module):
    """The model's OWN parameters behind a forward made of registered operators only: what torch DDP (or any tool that
    relies on autograd hooks) can wrap -- `DistributedDataParallel(FunctionalMLMModel(model))` reduces every gradient through
    its bucket hooks, where the module mirror's flat-buffer backward (and the drop-in built on it) must exchange gradients
    itself.  Parameters are shared with `model` (no copy), so optimizers / checkpoints of `model` keep working.
    forward(ids) -> {"loss", "n_labels"}; the device draws of the MLM mask and of the dropout masks advance exactly as in
    the module mirror (same seeds, same counters: a step here and a step there draw the same masks)."""

    def __init__(self, model):
        super().__init__()
        self.cfg = config_of(model)
        m = model.input_features.masking
        self.table = model.input_features.item_embedding_table.weight
        self.masked_emb = m.masked_item_embedding
        self.layers = torch.nn.ModuleList([torch.nn.ParameterList(p) for p in layer_params(model)])
        self._masking = [m]           # not sub-modules: only their seeds / counters are read and advanced
        self._transformer = [model.transformer_block.transformer]

    def forward(self, ids):
        m, t = self._masking[0], self._transformer[0]
        drop_offset = 0
        if self.training and self.cfg["dropout"] > 0:
            t._drop_offset += 1
            drop_offset = t._drop_offset
        loss, n = mlm_step(self.table, self.masked_emb, [list(p) for p in self.layers], self.cfg, ids, m.seed, m._rng_offset,
                           drop_seed=t.seed, drop_offset=drop_offset, training=self.training)
        if self.training:
            m._rng_offset += ids.numel()
        return {"loss": loss, "n_labels": n}


# ---------------------------------------------------------------------------------------------- multi-feature sessions (C3)
def input_block_of(model):
    """-> (params, spec) of the model's input block for `session_step`: tables / soft embeddings / projection in the column
    order the module mirror concatenates them (sorted feature names).  Supported shape = BASELINE configs[1] and [2]:
    sequence categoricals + SoftEmbedding continuous features, concat, optional ReLU projection, MLM; anything else raises."""
    from .features import ContinuousFeatures

    f = model.input_features
    cat, cont = f.categorical_module, f.continuous_module
    if f._aggregation != "concat" and len(f._feature_order) > 1:
        raise NotImplementedError("functional path: concat aggregation only")
    if cat._post is not None or getattr(cat, "_pre", None) is not None:
        raise NotImplementedError("functional path: no pre / post transformations on the categorical module")
    if cont is not None and isinstance(cont, ContinuousFeatures):
        raise NotImplementedError("functional path: continuous features as SoftEmbedding features only")
    tables, soft, layout, dims, id_names, dense_names = [], [], [], [], [], []
    for name in f._feature_order:
        dims.append(int(f._dims[name]))
        if name in cat.embedding_tables:
            tables.append(cat.embedding_tables[name].weight)
            id_names.append(name)
            layout.append(len(tables))
        else:
            se = cont.embedding_tables[name]
            ln = cont.post.feature_layer_norm[name] if cont.post is not None else None
            soft.append(dict(proj_w=se.projection_layer.weight, proj_b=se.projection_layer.bias, table=se.embedding_table.weight,
                             ln_w=None if ln is None else ln.weight, ln_b=None if ln is None else ln.bias,
                             eps=1e-5 if ln is None else float(ln.eps)))
            dense_names.append(name)
            layout.append(-len(soft))
    proj = f.projection_module[0][0] if f.projection_module is not None else None
    params = dict(tables=tables, soft=soft, proj=None if proj is None else (proj.weight, proj.bias),
                  masked_emb=f._masking.masked_item_embedding)
    spec = dict(layout=layout, dims=dims, id_names=id_names, dense_names=dense_names, item=id_names.index(cat.item_id))
    return params, spec


def session_step(block, spec, layers, cfg: Dict, inputs: Dict[str, torch.Tensor], mask_seed: int, mask_offset: int,
                 drop_seed: int = 0, drop_offset: int = 0, training: bool = True, n_labels: int = None):
    """`mlm_step` for a multi-feature input block (BASELINE configs[2]): soft embeddings -> concatenating gather -> ReLU
    projection -> MLM mask -> XLNet (every dropout site) -> tied next-item head, registered operators only.
    block / spec: `input_block_of(model)`; inputs: {feature name: [B, L] tensor}."""
    ids = [inputs[n].contiguous() for n in spec["id_names"]]
    item_ids = ids[spec["item"]]
    table = block["tables"][spec["item"]]
    B, L = item_ids.shape
    p = float(cfg.get("dropout", 0.0)) if training else 0.0
    if training:
        mask, labels, pos, lab, n = T4R.mlm_targets(item_ids, cfg["mlm_probability"], mask_seed, mask_offset, cfg["padding_idx"])
    else:
        mask, labels, counts = ops.mask_targets(item_ids, ops.MLM_EVAL_LAST, cfg["padding_idx"])
        n, pos, lab = ops.compact_labels(labels, counts, cfg["padding_idx"])
    pad = int(cfg["padding_idx"])
    if len(spec["layout"]) == 1 and block["proj"] is None:
        x = T4R.seq_item_embedding(item_ids, table, mask, block["masked_emb"], ops.MASK_MLM, pad)
    else:
        dense = [T4R.soft_embedding(inputs[nm].contiguous().float(), s["proj_w"], s["proj_b"], s["table"], s["ln_w"], s["ln_b"], s["eps"])
                 for nm, s in zip(spec["dense_names"], block["soft"])]
        x = T4R.seq_concat(ids, list(block["tables"]), dense, list(spec["layout"]), list(spec["dims"]), pad)
        W = x.shape[-1]
        if block["proj"] is not None:
            x = T4R.linear_relu(x.view(B * L, W), block["proj"][0], block["proj"][1]).view(B, L, -1)
        x = T4R.apply_mask(x, mask, block["masked_emb"], ops.MASK_MLM)
    D = x.shape[-1]
    pe = relative_positional_encoding(L, D).to(x.device).contiguous()
    pos_b = None
    if p > 0:
        x = T4R.dropout(x, p, drop_seed, ops.dropout_ctr_hi(drop_offset, 255, ops.SITE_INPUT))
        pos_b = T4R.pos_emb_dropout(pe, B, p, drop_seed, drop_offset)
    h = x.reshape(B * L, D)
    for li, prm in enumerate(layers):
        h, _ws = T4R.xlnet_layer_fwd(h, pe, list(prm), B, L, cfg["n_head"], cfg["eps"], p, drop_seed, drop_offset, li, pos_b)
    if p > 0:
        h = T4R.dropout(h, p, drop_seed, ops.dropout_ctr_hi(drop_offset, 255, ops.SITE_FINAL))
    N = int(n.item()) if n_labels is None else int(n_labels)
    rows = T4R.gather_label_rows(h, pos, N)
    y = lab[:N].contiguous()
    if _head_split_form(rows, table, N):
        loss = T4R.next_item_head(rows, table, y, 1.0 / cfg["temperature"], cfg["label_smoothing"])[0]
    else:
        loss = T4R.linear_softmax_ce(rows, table, y, 1.0 / cfg["temperature"], cfg["label_smoothing"])[0]
    return loss, N


class FunctionalSessionModel(torch.nn.Module):
    """`FunctionalMLMModel` for any input block `input_block_of` takes (configs[1] AND the multi-feature configs[2]): the model's
    own parameters behind a forward of registered operators only -- what torch DDP wraps.  forward(inputs dict) -> {"loss",
    "n_labels"}; masks and dropout draws advance exactly as in the module mirror."""

    def __init__(self, model):
        super().__init__()
        self.cfg = config_of(model)
        self.block, self.spec = input_block_of(model)
        # registered so that DDP / optimizers see them (shared with `model`, no copies)
        self.tables = torch.nn.ParameterList(self.block["tables"])
        self.soft = torch.nn.ParameterList([q for s in self.block["soft"] for q in (s["proj_w"], s["proj_b"], s["table"], s["ln_w"], s["ln_b"])
                                            if q is not None])
        self.proj = torch.nn.ParameterList([] if self.block["proj"] is None else list(self.block["proj"]))
        self.masked_emb = self.block["masked_emb"]
        self.layers = torch.nn.ModuleList([torch.nn.ParameterList(p) for p in layer_params(model)])
        self._masking = [model.input_features.masking]
        self._transformer = [model.transformer_block.transformer]

    def forward(self, inputs):
        m, t = self._masking[0], self._transformer[0]
        drop_offset = 0
        if self.training and self.cfg["dropout"] > 0:
            t._drop_offset += 1
            drop_offset = t._drop_offset
        n_tok = inputs[self.spec["id_names"][self.spec["item"]]].numel()
        loss, n = session_step(self.block, self.spec, [list(p) for p in self.layers], self.cfg, inputs, m.seed, m._rng_offset,
                               drop_seed=t.seed, drop_offset=drop_offset, training=self.training)
        if self.training:
            m._rng_offset += n_tok
        return {"loss": loss, "n_labels": n}
