"""Input block: host-side mirror of the reference's
  SequenceEmbeddingFeatures / EmbeddingFeatures   transformers4rec/torch/features/{sequence.py:43-94, embedding.py:51-276}
  SoftEmbeddingFeatures / SoftEmbedding           features/embedding.py:280-410, 517-556
  TabularSequenceFeatures                         features/sequence.py:97-296
with the same constructor arguments for the hot path, attribute contract (`masking`,
`item_id`, `item_embedding_table`, `categorical_module.item_seq`, `to_merge[...]`,
`projection_module`) and state_dict names.  All arithmetic runs in
csrc/embedding.hip / gemm_f32.hip through one fused autograd function.
"""
import functools
import math
import os
from typing import Dict, Optional

import torch
from torch import nn

from . import ops
from .transformations import TabularDropout, TabularLayerNorm, parse_post, parse_pre
from .masking import MaskSequence, _grad_buf, parse_masking
from .schema import Tags, categorical_cardinalities

# table-gradient scatter: stable sort by row id (done in the forward pass: it depends on the ids only) + segmented sum,
# deterministic, one owner per row (csrc/embedding_sorted.hip).  The fp32-row-atomics scatter (ops.embedding_bwd) is
# still in the C ABI for callers that want it; the module path does not use it.


def _table_scatter(ctx, name, grad_rows, ids_f, tab, col, dim, padding_idx):
    """d table += gradient rows of one feature.  grad_rows [n_lookups * ids_div, W]"""
    ids_div = grad_rows.numel() // grad_rows.shape[-1] // ids_f.numel()
    sink = getattr(tab, "_t4r_sparse_sink", None)
    if sink is not None:            # data-parallel row-sparse exchange (distributed.SparseRowExchange)
        sink.add(tab, ids_f, grad_rows, col, dim, ids_div, padding_idx)
        return
    srt = ctx.sorted_ids.get(name)
    if srt is None:
        srt = ops.sort_ids(ids_f, tab.shape[0], padding_idx)
    ops.embedding_bwd_sorted(grad_rows, srt[0], srt[1], _grad_buf(tab), col, dim, ids_div)


def _sort_ids_forward(ids, rows, padding_idx):
    """the forward-pass sort (it feeds only the backward), on the caller's stream.  (Rounds 3-4 ran it on a side stream; it
    was worth nothing at BASELINE configs[1] -- 2.834 vs 2.847 ms per step -- and a training step must not drive more than
    FOUR streams: caller + two weight-gradient streams + the collective's, DESIGN.md section 6.  Removed in round 6.)"""
    return ops.sort_ids(ids, rows, padding_idx)


def _sort_ids_forward_multi(ids_list, rows_list, padding_idx):
    """_sort_ids_forward for the F tables of one input block as ONE device sort -> [(keys, perm)] per table"""
    return ops.sort_ids_multi(ids_list, rows_list, [padding_idx] * len(ids_list))


class EmbeddingTable(nn.Module):
    """nn.Embedding-shaped parameter holder (`weight`, num_embeddings, embedding_dim, padding_idx)."""

    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, std=0.05):
        super().__init__()
        self.num_embeddings, self.embedding_dim, self.padding_idx = num_embeddings, embedding_dim, padding_idx
        self.weight = nn.Parameter(torch.empty(num_embeddings, embedding_dim))
        # TableConfig default initializer normal_(0, 0.05) runs AFTER nn.Embedding's padding row
        # zero-fill (features/sequence.py:75-81; embedding.py:460-464): row 0 is random, not zero.
        nn.init.normal_(self.weight, mean=0.0, std=std)

    def extra_repr(self):
        return f"{self.num_embeddings}, {self.embedding_dim}, padding_idx={self.padding_idx}"


class _Linear(nn.Module):
    """torch.nn.Linear-shaped parameter holder with the same default init."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(in_features) if in_features > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)


class _LayerNormParams(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class SoftEmbedding(nn.Module):
    """reference features/embedding.py:517-556: `embedding_table` [K, D] + `projection_layer` Linear(1, K)"""

    def __init__(self, num_embeddings, embeddings_dim):
        super().__init__()
        self.embedding_table = EmbeddingTable(num_embeddings, embeddings_dim)
        self.projection_layer = _Linear(1, num_embeddings)


class SequenceEmbeddingFeatures(nn.Module):
    """Categorical sequence features; holds `embedding_tables` and the stateful `item_seq`."""

    def __init__(self, table_sizes: Dict[str, tuple], item_id: Optional[str] = None, padding_idx: int = 0,
                 pre=None, post=None):
        super().__init__()
        self.item_id = item_id
        self.padding_idx = padding_idx
        self.embedding_tables = nn.ModuleDict(
            {n: EmbeddingTable(v, d, padding_idx=padding_idx) for n, (v, d) in table_sizes.items()})
        self.item_seq = None
        # pre / post transformations (TabularModule, tabular/base.py:226-283); state-dict names as the
        # reference: `_post.<i>.feature_layer_norm.<feature>.{weight,bias}`
        self._pre = parse_pre(pre)
        self._post = parse_post(post, {n: d for n, (v, d) in table_sizes.items()})

    @property
    def pre(self):
        return self._pre

    @property
    def post(self):
        return self._post

    @property
    def item_embedding_table(self):
        assert self.item_id is not None
        return self.embedding_tables[self.item_id]

    def item_ids(self, inputs):
        return inputs[self.item_id]


class TableConfig:
    """features/embedding.py:416-460: vocabulary_size, dim, initializer, combiner ("mean" | "sum" | "sqrtn"), name"""

    def __init__(self, vocabulary_size: int, dim: int, initializer=None, combiner: str = "mean", name: Optional[str] = None):
        if not isinstance(vocabulary_size, int) or vocabulary_size < 1:
            raise ValueError("Invalid vocabulary_size {}.".format(vocabulary_size))
        if not isinstance(dim, int) or dim < 1:
            raise ValueError("Invalid dim {}.".format(dim))
        if combiner not in ("mean", "sum", "sqrtn"):
            raise ValueError("Invalid combiner {}".format(combiner))
        if initializer is not None and not callable(initializer):
            raise ValueError("initializer must be callable if specified.")
        self.vocabulary_size, self.dim, self.combiner, self.name = vocabulary_size, dim, combiner, name
        # features/embedding.py:460-464: no initializer means normal_(mean 0, std 0.05), always applied by
        # table_to_embedding_module -- bag tables start at std 0.05 too, not at EmbeddingBag's N(0, 1)
        # (a functools.partial, as the reference: a lambda here made torch.save(model) / pickle fail)
        self.initializer = functools.partial(nn.init.normal_, mean=0.0, std=0.05) if initializer is None else initializer

    def __repr__(self):
        return (f"TableConfig(vocabulary_size={self.vocabulary_size!r}, dim={self.dim!r}, "
                f"combiner={self.combiner!r}, name={self.name!r})")


class FeatureConfig:
    """features/embedding.py:391-413"""

    def __init__(self, table: TableConfig, max_sequence_length: int = 0, name: Optional[str] = None):
        self.table, self.max_sequence_length, self.name = table, max_sequence_length, name


class _BagTable(nn.Module):
    """EmbeddingBagWrapper-shaped parameter holder (`weight`, `mode`).  The TableConfig's initializer -- normal_(0, 0.05)
    when the user gave none (features/embedding.py:460-464) -- always runs after construction (:86-93); built without one
    (direct use), the same N(0, 0.05) default as `EmbeddingTable`."""

    def __init__(self, num_embeddings, embedding_dim, mode="mean", initializer=None):
        super().__init__()
        self.num_embeddings, self.embedding_dim, self.mode = num_embeddings, embedding_dim, mode
        self.weight = nn.Parameter(torch.empty(num_embeddings, embedding_dim))
        nn.init.normal_(self.weight, mean=0.0, std=0.05)
        if initializer is not None:
            initializer(self.weight)

    def extra_repr(self):
        return f"{self.num_embeddings}, {self.embedding_dim}, mode={self.mode!r}"


class _EmbeddingBagsFn(torch.autograd.Function):
    """all bag lookups of one EmbeddingFeatures call.  forward: `t4r_embedding_bag_fwd` per feature, into its own
    [B, dim] tensor or (concat) into its columns of ONE [B, W] buffer; backward: per-lookup gradient rows
    (`t4r_embedding_bag_bwd_rows`) summed into the table by the deterministic sorted scatter, or handed to the
    data-parallel row-sparse sink.  Table gradients go straight into `.grad`, as everywhere on this path."""

    @staticmethod
    def forward(ctx, anchor, tables, lookups, concat, err):
        # lookups: [(values int64, offsets int64 | None, combiner)] aligned with `tables`
        B = lookups[0][1].numel() if lookups[0][1] is not None else lookups[0][0].shape[0]
        cols, c = [], 0
        for t in tables:
            cols.append(c)
            c += t.shape[1] if concat else 0
        wide = torch.empty((B, c), device=anchor.device, dtype=torch.float32) if concat else None
        outs, saved = [], []
        for t, (vals, offs, comb), col in zip(tables, lookups, cols):
            vals = vals.contiguous()
            offs = None if offs is None else offs.contiguous()
            outs.append(ops.embedding_bag_fwd(t.detach(), vals, offs, comb, out=wide, col=col, err_flag=err))
            saved.append((vals, offs, comb, 0 if offs is not None else vals.numel() // max(vals.shape[0], 1), col))
        ctx.tables, ctx.saved, ctx.concat = tables, saved, concat
        return (wide,) if concat else tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        for i, (tab, (vals, offs, comb, fixed_k, col)) in enumerate(zip(ctx.tables, ctx.saved)):
            dout = douts[0] if ctx.concat else douts[i]
            if dout is None:
                continue
            rows = ops.embedding_bag_bwd_rows(dout.contiguous(), vals.numel(), tab.shape[1], offs, fixed_k, comb, col=col)
            ids = vals.view(-1)
            sink = getattr(tab, "_t4r_sparse_sink", None)
            if sink is not None:
                sink.add_rows(tab, ids, rows, padding_idx=-1)
            else:
                ops.scatter_rows_sorted(_grad_buf(tab), ids, rows, padding_idx=-1)
        return None, None, None, None, None


class EmbeddingFeatures(nn.Module):
    """Drop-in for tr.EmbeddingFeatures (features/embedding.py:43-249), the NON-sequential categorical module whose
    tables are EmbeddingBags: per feature `inputs[name]` is

      * int64 [B]              -> bags of one id                       -> [B, dim]
      * int64 [B, K]           -> one bag of K ids per row (id 0 counts) -> [B, dim]
      * (values, offsets)      -> ragged bags; values [n] or [n, 1], offsets [B] or [B, 1] (the Merlin loader's
                                  sparse form, :229-236)                 -> [B, dim]

    combined with the table's `combiner`.  ("sqrtn" = sum / sqrt(n) is what TableConfig documents; the reference's
    torch.nn.EmbeddingBag rejects that mode at run time, and its tuple branch raises a TypeError as shipped -- the
    fixtures tests/golden/embedding_bag_*.npz record what the reference does compute.)
    forward returns {name: [B, dim]} (aggregation=None) or their concatenation in sorted-name order
    (aggregation="concat", tabular/aggregation.py:35-47).  state_dict names: `embedding_tables.<name>.weight`."""

    def __init__(self, feature_config: Dict[str, FeatureConfig], item_id: Optional[str] = None, pre=None, post=None,
                 aggregation: Optional[str] = None, schema=None):
        super().__init__()
        if pre is not None or post is not None:
            raise NotImplementedError("pre / post transformations on EmbeddingFeatures are off the hot path")
        if aggregation not in (None, "concat"):
            raise NotImplementedError("EmbeddingFeatures on the HIP path aggregates with None or 'concat'")
        self.item_id, self.feature_config, self.aggregation, self.schema = item_id, feature_config, aggregation, schema
        tables = {}
        for name, feature in feature_config.items():
            t = feature.table
            if name not in tables:
                tables[name] = _BagTable(t.vocabulary_size, t.dim, t.combiner, t.initializer)
        self.embedding_tables = nn.ModuleDict(tables)
        self.item_seq = None
        self._err = None

    @classmethod
    def from_schema(cls, schema, embedding_dims=None, embedding_dim_default: int = 64, infer_embedding_sizes=False,
                    infer_embedding_sizes_multiplier: float = 2.0, embeddings_initializers=None,
                    combiner: str = "mean", tags=None, item_id=None, automatic_build=True,
                    max_sequence_length=None, aggregation=None, pre=None, post=None, **kwargs):
        """features/embedding.py:95-221"""
        if infer_embedding_sizes:
            raise NotImplementedError("infer_embedding_sizes is off the hot path: pass embedding_dims")
        if tags:
            schema = schema.select_by_tag(tags)
        ids = schema.select_by_tag(Tags.ITEM_ID)
        if not item_id and len(ids) > 0:
            if len(ids) > 1:
                raise ValueError("Multiple columns with tag ITEM_ID found. Please specify the item_id column name.")
            item_id = ids.column_names[0]
        embedding_dims = embedding_dims or {}
        inits = embeddings_initializers or {}
        cfg = {name: FeatureConfig(TableConfig(card, embedding_dims.get(name, embedding_dim_default),
                                               initializer=inits.get(name), combiner=combiner, name=name))
               for name, card in categorical_cardinalities(schema).items()}
        if not cfg:
            return None
        return cls(cfg, item_id=item_id, pre=pre, post=post, aggregation=aggregation, schema=schema)

    @property
    def item_embedding_table(self):
        assert self.item_id is not None
        return self.embedding_tables[self.item_id]

    def item_ids(self, inputs):
        return inputs[self.item_id]

    def check_ids(self):
        """raises if a lookup since the last call saw an id outside its table (one host sync)"""
        if self._err is not None and int(self._err.item()) != 0:
            self._err.zero_()
            raise IndexError("EmbeddingFeatures: id outside [0, vocabulary_size)")

    def forward(self, inputs, **kwargs):
        names = [n for n in (sorted(self.feature_config) if self.aggregation == "concat" else self.feature_config)
                 if n in inputs]
        if not names:
            return {}
        tables, lookups = [], []
        for name in names:
            val = inputs[name]
            tab = self.embedding_tables[name]
            if isinstance(val, tuple):
                values, offsets = val
                values = values.reshape(-1)
                offsets = offsets[:, 0] if offsets.ndim == 2 else offsets
            else:
                values, offsets = val, None
                if val.ndim not in (1, 2):
                    raise ValueError(f"EmbeddingFeatures: {name} must be [B], [B, K] or a (values, offsets) tuple")
            tables.append(tab.weight)
            lookups.append((values, offsets, tab.mode))
        dev = tables[0].device
        if self._err is None or self._err.device != dev:
            self._err = torch.zeros(1, dtype=torch.int32, device=dev)
        concat = self.aggregation == "concat"
        outs = _EmbeddingBagsFn.apply(tables[0], tables, lookups, concat, self._err)
        if self.item_id:
            self.item_seq = self.item_ids(inputs)
        return outs[0] if concat else dict(zip(names, outs))

    def forward_output_size(self, input_sizes=None):
        return {name: torch.Size([-1, f.table.dim]) for name, f in self.feature_config.items()}


class _FeaturePost(nn.Module):
    def __init__(self, dims: Dict[str, int]):
        super().__init__()
        self.feature_layer_norm = nn.ModuleDict({n: _LayerNormParams(d) for n, d in dims.items()})


class SoftEmbeddingFeatures(nn.Module):
    """Continuous features through SoftEmbedding (+ per-feature LayerNorm in `post`)."""

    def __init__(self, table_sizes: Dict[str, tuple], layer_norm: bool = True, pre=None, post=None):
        super().__init__()
        if post is not None and not layer_norm:
            raise NotImplementedError("post transformations on soft embeddings other than the built-in "
                                      "layer norm are off the hot path")
        self._pre = parse_pre(pre)   # `post` is replaced by the LayerNorm when layer_norm=True (embedding.py:306-309)
        self.embedding_tables = nn.ModuleDict({n: SoftEmbedding(k, d) for n, (k, d) in table_sizes.items()})
        self.post = _FeaturePost({n: d for n, (k, d) in table_sizes.items()}) if layer_norm else None


class ContinuousFeatures(nn.Module):
    """Continuous sequence features passed through as width-1 columns
    (features/continuous.py:60-63: unsqueeze(-1)); no parameters."""

    def __init__(self, names, pre=None, post=None):
        super().__init__()
        self.names = list(names)
        self._pre = parse_pre(pre)
        self._post = parse_post(post, {n: 1 for n in self.names})   # LayerNorm skips dim-1 features
        self.embedding_tables = nn.ModuleDict()
        self.post = None


_POST_SALT = 4
_post_seed_value = None


def post_seed():
    """Philox key of the TabularDropout masks: rng.default_seed(_POST_SALT), resolved at the first draw"""
    global _post_seed_value
    if _post_seed_value is None:
        from .rng import default_seed

        _post_seed_value = default_seed(_POST_SALT)
    return _post_seed_value


def set_post_seed(value):
    """restores the TabularDropout key (rng.set_rng_state); None re-resolves it at the next draw"""
    global _post_seed_value
    _post_seed_value = None if value is None else int(value) & 0x7FFFFFFFFFFFFFFF


def _post_fwd(owner, post, name, fidx, e2d, step):
    """Applies the module's post transformations to one feature's rows e2d [rows, D].
    Dropout directly followed by the feature's LayerNorm runs as one fused launch.
    -> (rows after post, records for _post_bwd)"""
    saved = []
    mods = list(post)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, TabularDropout):
            p = m.dropout_rate if owner.training else 0.0
            ctr = ops.dropout_ctr_hi(step, 0xFE, fidx * 8 + i)
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(nxt, TabularLayerNorm) and name in nxt.feature_layer_norm:
                ln = nxt.feature_layer_norm[name]
                drop = (p, post_seed(), ctr) if p > 0 else ops.NO_DROP
                y, mean, rstd = ops.add_layernorm_fwd(e2d, None, ln.weight.detach(), ln.bias.detach(), ln.eps, drop)
                saved.append(("ln", e2d, mean, rstd, ln, drop))
                e2d = y
                i += 2
                continue
            if p > 0:
                e2d = ops.dropout(e2d, p, post_seed(), ctr).view(e2d.shape)
                saved.append(("drop", p, ctr))
        elif name in m.feature_layer_norm:
            ln = m.feature_layer_norm[name]
            y, mean, rstd = ops.add_layernorm_fwd(e2d, None, ln.weight.detach(), ln.bias.detach(), ln.eps)
            saved.append(("ln", e2d, mean, rstd, ln, ops.NO_DROP))
            e2d = y
        i += 1
    return e2d, saved


def _post_bwd(saved, dy):
    for rec in reversed(saved):
        if rec[0] == "drop":
            dy = ops.dropout(dy, rec[1], post_seed(), rec[2]).view(dy.shape)
        else:
            _, a, mean, rstd, ln, drop = rec
            r = ops.add_layernorm_bwd(a, None, ln.weight.detach(), mean, rstd, dy, _grad_buf(ln.weight),
                                      _grad_buf(ln.bias), drop=drop)
            dy = r[1] if drop[0] > 0 else r
    return dy


class _SeqFeaturesFn(torch.autograd.Function):
    """gather (+soft embeddings) -> aggregate -> [ReLU projection] -> masking, one autograd node.
    Parameter gradients are accumulated straight into `.grad` (flat-buffer friendly)."""

    @staticmethod
    def forward(ctx, anchor, mod, inputs, training, testing):
        ctx.carrier = bool(mod.__dict__.get("_t4r_carrier_on", False))      # anchor = the GradCarrier's one-float output
        cat, cont = mod.categorical_module, mod.continuous_module
        names = mod._feature_order
        item_ids = inputs[cat.item_id].contiguous()
        B, L = item_ids.shape
        masking = mod._masking
        mask_mode, mask, L_out = ops.MASK_NONE, None, L
        if masking is not None:
            masking.compute_masked_targets(item_ids, training=training, testing=testing)
            mask = masking.mask_schema
            mask_mode = masking.apply_mode(training, testing)
            L_out = mask.shape[1]
        feats, soft_saved, post_saved = [], {}, {}
        step = mod._post_step
        for fidx, name in enumerate(names):
            col, dim = mod._cols[name], mod._dims[name]
            if name in cat.embedding_tables:
                tab = cat.embedding_tables[name].weight
                ids_f = inputs[name].contiguous()
                # a [B] id tensor is a non-sequential (context) feature: looked up once per session
                # and broadcast over L (reference: tabular/base.py:53-63)
                per_session = ids_f.ndim == 1
                f = dict(kind=2 if per_session else 0, input=ids_f, table=tab.detach(), dim=dim, col=col,
                         rows=tab.shape[0])
                if cat._post is not None:
                    # post transformations act on the feature's own rows before the aggregation: gather
                    # them densely, transform, and hand them to the aggregation as dense rows
                    Lf = 1 if per_session else L
                    e = ops.seq_features_fwd([dict(f, kind=0, col=0)], "concat", B, Lf, Lf, dim,
                                             err_flag=mod._err_flag(item_ids.device)).view(B * Lf, dim)
                    e, post_saved[name] = _post_fwd(cat, cat._post, name, fidx, e, step)
                    f = dict(kind=3 if per_session else 1, input=e, table=None, dim=dim, col=col)
                feats.append(f)
            elif isinstance(cont, ContinuousFeatures):
                x = inputs[name].contiguous().float().view(B * L, 1)
                if cont._post is not None:
                    x, post_saved[name] = _post_fwd(cont, cont._post, name, fidx, x, step)
                feats.append(dict(kind=1, input=x, table=None, dim=1, col=col))
            else:
                se = cont.embedding_tables[name]
                ln = cont.post.feature_layer_norm[name] if cont.post is not None else None
                x = inputs[name].contiguous().float()
                rows = ops.soft_embedding_fwd(
                    x, se.projection_layer.weight.detach(), se.projection_layer.bias.detach(),
                    se.embedding_table.weight.detach(), None if ln is None else ln.weight.detach(),
                    None if ln is None else ln.bias.detach(), 1e-5 if ln is None else ln.eps)
                soft_saved[name] = x
                feats.append(dict(kind=1, input=rows, table=None, dim=dim, col=col))
        W = mod._agg_width
        proj = mod.projection_module
        fuse_mask = masking is not None and proj is None
        memb = masking.masked_item_embedding if masking is not None else None
        item_feat = names.index(cat.item_id) if cat.item_id in names else -1
        agg_out = ops.seq_features_fwd(
            feats, mod._aggregation, B, L, L_out, W, item_feat=item_feat,
            mask_mode=mask_mode if fuse_mask else ops.MASK_NONE, mask=mask if fuse_mask else None,
            masked_emb=memb.detach() if fuse_mask else None, err_flag=mod._err_flag(item_ids.device))
        out = agg_out
        if proj is not None:
            lin = proj[0][0]
            out = ops.gemm(agg_out.view(B * L_out, W), lin.weight.detach(), False, True,
                           bias=lin.bias.detach(), epilogue=ops.EPI_BIAS_RELU).view(B, L_out, -1)
            if masking is not None:
                ops.apply_mask_fwd_(out, mask, memb.detach(), mask_mode)
        ctx.mod, ctx.inputs, ctx.soft_saved, ctx.post_saved = mod, inputs, soft_saved, post_saved
        # the sort behind the deterministic table gradient depends on the ids only: done here, in the forward
        ctx.sorted_ids = {}
        if training:
            todo = []
            for name in names:
                if name in cat.embedding_tables:
                    tab = cat.embedding_tables[name].weight
                    if tab.requires_grad and getattr(tab, "_t4r_sparse_sink", None) is None:
                        todo.append((name, inputs[name].contiguous(), tab.shape[0]))
            if len(todo) > 1 and len({t[1].numel() for t in todo}) == 1 and len(todo) <= 16:
                # every table of a multi-feature block in ONE device sort (a third of the launches: ops.sort_ids_multi)
                srt = _sort_ids_forward_multi([t[1] for t in todo], [t[2] for t in todo], cat.padding_idx)
                for (name, _i, _r), s_f in zip(todo, srt):
                    ctx.sorted_ids[name] = s_f
            else:
                for name, ids_n, rows_n in todo:
                    ctx.sorted_ids[name] = _sort_ids_forward(ids_n, rows_n, cat.padding_idx)
        ctx.mask_mode, ctx.mask, ctx.dims, ctx.post_step = mask_mode, mask, (B, L, L_out, W), step
        ctx.agg_out = agg_out if proj is not None else None
        ctx.proj_out = out if proj is not None else None
        return out

    @staticmethod
    def backward(ctx, dy):
        mod = ctx.mod
        cat, cont = mod.categorical_module, mod.continuous_module
        B, L, L_out, W = ctx.dims
        if L_out != L:
            raise RuntimeError("the MLM inference grid (L+1) is forward-only")
        masking, proj = mod._masking, mod.projection_module
        if masking is not None:     # out of place: the incoming gradient is not ours to write, and a clone is a 10 MB copy
            d = ops.apply_mask_bwd(dy, ctx.mask, _grad_buf(masking.masked_item_embedding), ctx.mask_mode)
        else:
            d = dy.contiguous().clone()
        if proj is not None:
            lin = proj[0][0]
            H = d.shape[-1]
            d2 = d.view(B * L, H)
            ops.act_bwd_bias(d2, ctx.proj_out.view(B * L, H), None if lin.bias is None else _grad_buf(lin.bias), 1)
            # the projection's weight gradient (K = tokens): split-K with the partials added in split order, not atomics --
            # it was the one order-dependent sum of a configs[2] step
            ops.gemm_wgrad(d2, ctx.agg_out.view(B * L, W), _grad_buf(lin.weight))
            d = ops.gemm(d2, lin.weight.detach(), False, False).view(B, L, W)
        names = mod._feature_order
        agg = mod._aggregation
        if agg == "element-wise-sum-item-multi":
            item = cat.item_id
            def again(n):   # the feature as the forward aggregated it (post-transformed rows are recomputed)
                ids_n = ctx.inputs[n].contiguous()
                tabn = cat.embedding_tables[n].weight.detach()
                f = dict(kind=2 if ids_n.ndim == 1 else 0, input=ids_n, table=tabn, dim=W, col=0, rows=tabn.shape[0])
                if n in ctx.post_saved:
                    Lf = 1 if ids_n.ndim == 1 else L
                    e = ops.seq_features_fwd([dict(f, kind=0)], "concat", B, Lf, Lf, W).view(B * Lf, W)
                    e, _ = _post_fwd(cat, cat._post, n, names.index(n), e, ctx.post_step)
                    f = dict(kind=3 if ids_n.ndim == 1 else 1, input=e, table=None, dim=W, col=0)
                return f

            f_item = [again(item)]
            f_other = [again(n) for n in names if n != item]
            e_item = ops.seq_features_fwd(f_item, "element-wise-sum", B, L, L, W)
            e_other = ops.seq_features_fwd(f_other, "element-wise-sum", B, L, L, W)
            d_item, d_other = ops.mul(d, e_other), ops.mul(d, e_item)
        for name in names:
            col, dim = mod._cols[name], mod._dims[name]
            if name in cat.embedding_tables:
                tab = cat.embedding_tables[name].weight
                if not tab.requires_grad:
                    continue
                src = d
                if agg == "element-wise-sum-item-multi":
                    src = d_item if name == cat.item_id else d_other
                ids_f = ctx.inputs[name].contiguous()
                if name in ctx.post_saved:
                    src2 = src.view(B * L, -1)
                    if ids_f.ndim == 1:     # broadcast over L: the gradient is the sum over the sequence
                        g = ops.seq_sum_cols(src2, col, dim, B, L)
                    else:
                        g = ops.copy_cols_out(src2, col, dim) if src2.shape[1] != dim else src2
                    g = _post_bwd(ctx.post_saved[name], g)
                    _table_scatter(ctx, name, g, ids_f, tab, 0, dim, cat.padding_idx)
                else:
                    _table_scatter(ctx, name, src.view(-1, src.shape[-1]), ids_f, tab, col, dim, cat.padding_idx)
            elif isinstance(cont, ContinuousFeatures):
                continue    # no parameters behind a pass-through column
            else:
                se = cont.embedding_tables[name]
                ln = cont.post.feature_layer_norm[name] if cont.post is not None else None
                ops.soft_embedding_bwd(
                    d, ctx.soft_saved[name], se.projection_layer.weight.detach(),
                    se.projection_layer.bias.detach(), se.embedding_table.weight.detach(),
                    None if ln is None else ln.weight.detach(), _grad_buf(se.projection_layer.weight),
                    _grad_buf(se.projection_layer.bias), _grad_buf(se.embedding_table.weight),
                    None if ln is None else _grad_buf(ln.weight), None if ln is None else _grad_buf(ln.bias),
                    col, 1e-5 if ln is None else ln.eps)
        # the anchor's slot: a defined (zero) gradient for the GradCarrier's output, so that its node -- the last of the
        # backward pass -- runs and hands the carried parameter gradients to autograd
        return (torch.zeros(1, device=dy.device) if ctx.carrier else None), None, None, None, None


class TabularSequenceFeatures(nn.Module):
    """Drop-in for tr.TabularSequenceFeatures on the hot path (features/sequence.py:97-296).

    forward(inputs: Dict[str, Tensor[B, L]], training=False, testing=False) -> Tensor[B, L(+1), H]
    """

    def __init__(self, categorical_module: SequenceEmbeddingFeatures,
                 continuous_module: Optional[SoftEmbeddingFeatures] = None, aggregation: str = "concat",
                 projection_dim: Optional[int] = None, masking: Optional[MaskSequence] = None,
                 schema=None, max_sequence_length: Optional[int] = None):
        super().__init__()
        if aggregation not in ops.AGG:
            raise ValueError(f"aggregation must be one of {sorted(ops.AGG)}")
        self.schema = schema
        self.max_sequence_length = max_sequence_length
        mods = {}
        if continuous_module is not None:
            mods["continuous_module"] = continuous_module
        mods["categorical_module"] = categorical_module
        self.to_merge = nn.ModuleDict(mods)
        self._aggregation = aggregation
        dims = {n: t.embedding_dim for n, t in categorical_module.embedding_tables.items()}
        if isinstance(continuous_module, ContinuousFeatures):
            dims.update({n: 1 for n in continuous_module.names})
        elif continuous_module is not None:
            dims.update({n: s.embedding_table.embedding_dim for n, s in continuous_module.embedding_tables.items()})
        # ConcatFeatures order = sorted(feature names) (tabular/aggregation.py:43)
        self._feature_order = sorted(dims)
        self._dims = dims
        if aggregation == "concat":
            off, cols = 0, {}
            for n in self._feature_order:
                cols[n] = off
                off += dims[n]
            self._cols, self._agg_width = cols, off
        else:
            if len(set(dims.values())) != 1:
                raise ValueError(f"All features must have the same dimension for element-wise aggregation: {dims}")
            self._cols, self._agg_width = {n: 0 for n in dims}, next(iter(dims.values()))
        self.projection_module = None
        if projection_dim:
            # MLPBlock([d_output]) -> SequentialBlock(DenseBlock(Linear, ReLU)) (block/mlp.py:68-143)
            self.projection_module = nn.Sequential(nn.Sequential(_Linear(self._agg_width, projection_dim)))
        self._masking = None
        self.set_masking(masking)
        self._err = None
        self._post_step = 0     # Philox stream position of the post-dropout / swap-noise draws

    # ------------------------------------------------------------------ reference attribute contract
    @property
    def categorical_module(self):
        return self.to_merge["categorical_module"]

    @property
    def continuous_module(self):
        return self.to_merge["continuous_module"] if "continuous_module" in self.to_merge else None

    @property
    def masking(self):
        return self._masking

    def set_masking(self, value):
        self._masking = value

    @property
    def item_id(self):
        return self.categorical_module.item_id

    @property
    def item_embedding_table(self):
        return self.categorical_module.item_embedding_table

    @property
    def aggregation(self):
        return self._aggregation

    def output_size(self, input_size=None):
        H = self.projection_module[0][0].out_features if self.projection_module is not None else self._agg_width
        return torch.Size([-1, self.max_sequence_length or -1, H])

    forward_output_size = output_size

    def _err_flag(self, device):
        if self._err is None or self._err.device != device:
            self._err = torch.zeros(1, dtype=torch.int32, device=device)
        return self._err

    def check_ids(self):
        """Raises if any lookup id was out of range since the last check (device flag, syncs)."""
        if self._err is not None and int(self._err.item()) != 0:
            self._err.zero_()
            raise IndexError("embedding lookup id out of range")

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_schema(cls, schema, continuous_tags=(Tags.CONTINUOUS,), categorical_tags=(Tags.CATEGORICAL,),
                    aggregation=None, max_sequence_length=None, continuous_soft_embeddings=False,
                    d_output=None, masking=None, embedding_dims=None, embedding_dim_default=64,
                    soft_embedding_cardinalities=None, soft_embedding_cardinality_default=10,
                    soft_embedding_dims=None, soft_embedding_dim_default=8, layer_norm=True,
                    projection=None, continuous_projection=None, pre=None, post=None, **kwargs):
        """Same keyword surface as the reference (features/sequence.py:140-229,
        features/embedding.py:103-221, :313-410) for the arguments the hot path uses."""
        if projection is not None or continuous_projection is not None:
            raise NotImplementedError("only d_output (ReLU MLP) projection is on the hot path")
        cat_schema = schema.select_by_tag(list(categorical_tags))
        cards = categorical_cardinalities(cat_schema)
        if not cards:
            raise ValueError("a categorical_module including an item_id column is required")
        item_cols = schema.select_by_tag(Tags.ITEM_ID).column_names
        item_id = item_cols[0] if item_cols else None
        embedding_dims = embedding_dims or {}
        tables = {n: (v, embedding_dims.get(n, embedding_dim_default)) for n, v in cards.items()}
        # pre / post reach every feature module through **kwargs in the reference
        # (features/tabular.py:150-185); soft embeddings replace `post` by their LayerNorm
        cat = SequenceEmbeddingFeatures(tables, item_id=item_id, pre=pre, post=post)
        cont = None
        cont_names = [c for c in schema.select_by_tag(list(continuous_tags)).column_names if c not in cards]
        if cont_names and not continuous_soft_embeddings:
            cont = ContinuousFeatures(cont_names, pre=pre, post=post)     # features/continuous.py:60-63
        elif cont_names:
            sc, sd = soft_embedding_cardinalities or {}, soft_embedding_dims or {}
            cont = SoftEmbeddingFeatures(
                {n: (sc.get(n, soft_embedding_cardinality_default), sd.get(n, soft_embedding_dim_default))
                 for n in cont_names}, layer_norm=layer_norm, pre=pre, post=None if layer_norm else post)
        if (masking or d_output) and not aggregation:
            aggregation = "concat"
        out = cls(cat, cont, aggregation or "concat", d_output, None, schema, max_sequence_length)
        hidden = out.output_size()[-1]
        if masking is not None and item_id is None:
            raise ValueError("For masking a categorical_module is required including an item_id.")
        out.set_masking(parse_masking(masking, hidden, **kwargs))
        return out

    # ------------------------------------------------------------------ forward
    def _apply_pre(self, inputs):
        """StochasticSwapNoise of each feature module on its own features (TabularModule.__call__ runs
        `pre` before forward, tabular/base.py:389); the padding mask comes from the ORIGINAL item ids."""
        active = {}
        out = inputs
        for mi, (mname, m) in enumerate(self.to_merge.items()):
            ssn = getattr(m, "_pre", None)
            if ssn is None or not ssn.training:
                continue
            if out is inputs:
                out = dict(inputs)
                item_ids = inputs[self.categorical_module.item_id].contiguous()
            names = list(m.names) if isinstance(m, ContinuousFeatures) else list(m.embedding_tables.keys())
            out.update(ssn.augment_module(inputs, [n for n in names if n in inputs], item_ids, mname, mi))
            active[id(ssn)] = ssn
        for ssn in active.values():     # one stream position per forward, also when modules share the object
            ssn.next_step()
        return out

    def forward(self, inputs, training=False, testing=False, **kwargs):
        cat = self.categorical_module
        inputs = self._apply_pre(inputs)
        if cat.item_id:
            cat.item_seq = cat.item_ids(inputs)  # stateful, as the reference (embedding.py:242-245); after `pre`
        if self.training:
            self._post_step += 1
        anchor = cat.item_embedding_table.weight
        carried = self.__dict__.get("_t4r_carrier_params")
        object.__setattr__(self, "_t4r_carrier_on", False)
        if carried and torch.is_grad_enabled() and (training or self.training):
            from .masking import GradCarrier

            GradCarrier.release(carried)                 # leftovers of a forward whose backward never ran
            anchor = GradCarrier.apply(carried, *carried)      # autograd-visible parameter gradients (masking.GradCarrier)
            object.__setattr__(self, "_t4r_carrier_on", True)
        return _SeqFeaturesFn.apply(anchor, self, inputs, training, testing)
