"""Executable drop-in: subclasses of the REFERENCE's three hot-path modules whose forward runs on
the HIP path (north_star "drops into transformers4rec.torch as a replacement backbone").

    import transformers4rec.torch as tr
    from transformers4rec_amd import dropin
    hip = dropin.install(tr)              # tr.TabularSequenceFeatures / TransformerBlock / NextItemPredictionTask
    model = tr.XLNetConfig.build(...).to_torch_model(inputs, task)   # built by the reference's own code
    # or, for a model that already exists (e.g. loaded from a checkpoint):
    dropin.convert_model(model)           # swaps the classes of the three modules in place

The subclasses add NO parameters, buffers or sub-modules: construction (`from_schema`, `build`,
`to_torch_model`), the module tree, `state_dict` names, `isinstance` gates
(block/base.py:149-154, config/transformer.py:113, model/base.py:410) are the reference's own.
Only `forward` is replaced: it drives a *shadow* -- the host mirror of this package
(features.TabularSequenceFeatures, transformer.TransformerBlock, prediction_task.
NextItemPredictionTask) whose nn.Parameters ARE the reference module's parameters (tied by
state_dict name, same objects, no copy) -- so gradients land in the reference parameters' `.grad`
and an optimizer built on `model.parameters()` trains as before.  The shadow is built lazily on
the first forward and is not registered as a sub-module.

Data parallelism: the HIP backward writes parameter gradients straight into `.grad` (the autograd
functions return None for parameters), so AccumulateGrad hooks never fire and torch DDP -- what HF
Trainer wraps the reference model in (torch/trainer.py:131-161) -- would never see them: with
`find_unused_parameters=True` every HIP parameter is marked unused and the replicas diverge silently.
A training forward under `torch.distributed` world_size > 1 therefore RAISES until the caller has
said -- BEFORE that forward -- that it exchanges `.grad` itself after backward:
`convert_model(model, data_parallel=True)` / `enable_data_parallel(model)` (per model), then
`sync_gradients(model)` (one flat all-reduce of every `.grad`, averaged: DDP's semantics) between
`backward()` and `optimizer.step()`, or `distributed.GradReducer` / `SparseRowExchange`.

There is no CPU fallback: a CPU tensor raises _lib.T4RHipError like the rest of the package.
Configurations off the hot path (PLM/RTD masking, custom projection blocks, pretrained-embedding
modules, transformer bodies other than XLNet / GPT-2 / BERT) raise NotImplementedError when the
shadow is built; nothing silently routes through the reference's torch code.
"""
import torch
from torch import nn

from . import features as F
from . import masking as M
from . import prediction_task as P
from . import transformations as T
from . import transformer as X
from . import transformer_hf as H

_SHADOW = "_t4r_hip_shadow"


# ------------------------------------------------------------------------------------------ tying
def _walk(mod, dotted):
    for part in dotted:
        mod = getattr(mod, part)
    return mod


def tie_by_name(shadow: nn.Module, ref: nn.Module, what="module"):
    """Makes every parameter / buffer of `shadow` the SAME object as the equally named one of `ref`.
    Raises if a shadow parameter has no reference counterpart or the shapes differ."""
    ref_params = dict(ref.named_parameters(remove_duplicate=False))
    ref_bufs = dict(ref.named_buffers(remove_duplicate=False))
    missing = []
    for name, p in list(shadow.named_parameters(remove_duplicate=False)):
        src = ref_params.get(name)
        if src is None:
            missing.append(name)
            continue
        if tuple(src.shape) != tuple(p.shape):
            raise ValueError(f"dropin: {what} parameter {name}: shape {tuple(src.shape)} in the reference, "
                             f"{tuple(p.shape)} on the HIP side")
        path = name.split(".")
        _walk(shadow, path[:-1])._parameters[path[-1]] = src
    for name, b in list(shadow.named_buffers(remove_duplicate=False)):
        src = ref_bufs.get(name)
        if src is None:
            missing.append(name)
            continue
        path = name.split(".")
        _walk(shadow, path[:-1])._buffers[path[-1]] = src
    if missing:
        raise NotImplementedError(f"dropin: the reference {what} has no tensor named {missing}; "
                                  "this configuration is off the HIP hot path")
    return shadow


def _cls(obj):
    return type(obj).__name__


def _meta():
    return torch.device("meta")


# ------------------------------------------------------------------------------------------ masking
def shadow_masking(ref_masking):
    if ref_masking is None:
        return None
    name = _cls(ref_masking)
    kw = dict(hidden_size=int(ref_masking.masked_item_embedding.shape[0]),
              padding_idx=ref_masking.padding_idx,
              eval_on_last_item_seq_only=getattr(ref_masking, "eval_on_last_item_seq_only", True))
    if name == "MaskedLanguageModeling":
        m = M.MaskedLanguageModeling(mlm_probability=ref_masking.mlm_probability, **kw)
    elif name == "CausalLanguageModeling":
        m = M.CausalLanguageModeling(
            train_on_last_item_seq_only=getattr(ref_masking, "train_on_last_item_seq_only", False), **kw)
    else:
        raise NotImplementedError(f"dropin: masking {name} is off the HIP hot path (MLM / CLM only)")
    m._parameters["masked_item_embedding"] = ref_masking.masked_item_embedding
    return m


# ------------------------------------------------------------------------------------------ input block
def _transformations(tfm):
    """reference SequentialTabularTransformations / list / single module / None -> flat list"""
    if tfm is None:
        return []
    if isinstance(tfm, (nn.ModuleList, nn.Sequential, list, tuple)):
        out = []
        for t in tfm:
            out += _transformations(t)
        return out
    return [tfm]


def _map_pre(tfm):
    items = _transformations(tfm)
    if not items:
        return None
    out = []
    for t in items:
        if _cls(t) != "StochasticSwapNoise":
            raise NotImplementedError(f"dropin: pre transformation {_cls(t)} is off the HIP hot path")
        out.append(T.StochasticSwapNoise(schema=getattr(t, "schema", None), pad_token=t.pad_token,
                                         replacement_prob=t.replacement_prob))
    return out


def _map_post(tfm):
    items = _transformations(tfm)
    if not items:
        return None
    out = []
    for t in items:
        if _cls(t) == "TabularDropout":
            rate = t.dropout_rate if hasattr(t, "dropout_rate") else t.dropout.p
            out.append(T.TabularDropout(rate))
        elif _cls(t) == "TabularLayerNorm":
            out.append(T.TabularLayerNorm({n: int(ln.weight.shape[0]) for n, ln in t.feature_layer_norm.items()}))
        else:
            raise NotImplementedError(f"dropin: post transformation {_cls(t)} is off the HIP hot path")
    return out


def _aggregation_name(agg):
    if agg is None or isinstance(agg, str):
        return agg or "concat"
    name = {"ConcatFeatures": "concat", "ElementwiseSum": "element-wise-sum",
            "ElementwiseSumItemMulti": "element-wise-sum-item-multi"}.get(_cls(agg))
    if name is None:
        raise NotImplementedError(f"dropin: aggregation {_cls(agg)} is off the HIP hot path")
    return name


def _dense_relu_dim(proj):
    """MLPBlock([d]) = SequentialBlock(DenseBlock(Linear, ReLU)) (block/mlp.py:68-143) -> d, or raises"""
    if proj is None:
        return None
    try:
        blocks = list(proj)
        dense = list(blocks[0])
        lin = dense[0]
        ok = (len(blocks) == 1 and hasattr(lin, "weight") and lin.weight.ndim == 2
              and (len(dense) == 1 or (len(dense) == 2 and _cls(dense[1]) == "ReLU")))
    except TypeError:
        ok = False
    if not ok:
        raise NotImplementedError("dropin: only the d_output projection MLPBlock([d]) (one Linear + ReLU) is on "
                                  "the HIP hot path")
    return int(lin.weight.shape[0])


def shadow_features(ref):
    """HIP-side TabularSequenceFeatures sharing every parameter with the reference module `ref`."""
    merge = ref.to_merge
    def _empty(m):      # e.g. a PretrainedEmbeddingFeatures the schema selected no column for
        inc = getattr(getattr(m, "filter_features", None), "to_include", None)
        return inc is not None and len(inc) == 0 and not list(m.parameters())

    extra = [k for k in merge.keys() if k not in ("categorical_module", "continuous_module") and not _empty(merge[k])]
    if extra or "categorical_module" not in merge:
        raise NotImplementedError(f"dropin: feature modules {extra or 'without categorical_module'} are off the "
                                  "HIP hot path")
    rc = merge["categorical_module"]
    with _meta():
        tables = {n: (int(t.weight.shape[0]), int(t.weight.shape[1])) for n, t in rc.embedding_tables.items()}
        cat = F.SequenceEmbeddingFeatures(tables, item_id=rc.item_id, padding_idx=getattr(rc, "padding_idx", 0),
                                          pre=_map_pre(getattr(rc, "pre", None)),
                                          post=_map_post(getattr(rc, "post", None)))
        cont = None
        if "continuous_module" in merge:
            rk = merge["continuous_module"]
            kind = _cls(rk)
            if kind == "SoftEmbeddingFeatures":
                soft = {n: (int(s.embedding_table.weight.shape[0]), int(s.embedding_table.weight.shape[1]))
                        for n, s in rk.embedding_tables.items()}
                post = getattr(rk, "post", None)
                has_ln = post is not None and any(_cls(t) in ("TabularLayerNorm", "_FeaturePost")
                                                  for t in _transformations(post))
                cont = F.SoftEmbeddingFeatures(soft, layer_norm=has_ln, pre=_map_pre(getattr(rk, "pre", None)))
            elif kind == "ContinuousFeatures":
                names = list(rk.filter_features.to_include) if hasattr(rk, "filter_features") else list(rk.names)
                cont = F.ContinuousFeatures(names, pre=_map_pre(getattr(rk, "pre", None)),
                                            post=_map_post(getattr(rk, "post", None)))
            else:
                raise NotImplementedError(f"dropin: continuous module {kind} (e.g. continuous_projection) is off "
                                          "the HIP hot path")
        sh = F.TabularSequenceFeatures(cat, cont, _aggregation_name(ref.aggregation),
                                       _dense_relu_dim(ref.projection_module), None, getattr(ref, "schema", None),
                                       getattr(ref, "max_sequence_length", None))
        sh.set_masking(shadow_masking(ref.masking))
    tie_by_name(sh, ref, "TabularSequenceFeatures")
    return sh


# ------------------------------------------------------------------------------------------ transformer
def shadow_block(ref):
    """HIP-side TransformerBlock whose body shares every parameter with `ref.transformer` (an HF
    XLNetModel / GPT2Model / BertModel, or this package's mirror of one)."""
    body = ref.transformer
    c = body.config
    kind = _cls(body)
    with _meta():
        if kind == "XLNetModel":
            if getattr(c, "attn_type", "bi") != "bi" or getattr(c, "bi_data", False) or \
                    getattr(c, "clamp_len", -1) != -1 or getattr(c, "same_length", False):
                raise NotImplementedError("dropin: XLNet with attn_type != 'bi' / bi_data / clamp_len / same_length "
                                          "is off the HIP hot path")
            if getattr(c, "ff_activation", "gelu") != "gelu":
                raise NotImplementedError("dropin: XLNet ff_activation must be 'gelu'")
            cfg = X.XLNetConfig.build(d_model=c.d_model, n_head=c.n_head, n_layer=c.n_layer,
                                      initializer_range=c.initializer_range, layer_norm_eps=c.layer_norm_eps,
                                      dropout=c.dropout, mem_len=getattr(c, "mem_len", 1) or 1)
            cfg.d_inner, cfg.vocab_size = c.d_inner, c.vocab_size
            model = X.XLNetModel(cfg)
        elif kind == "GPT2Model":
            act = getattr(c, "activation_function", "gelu")
            if act != "gelu":
                raise NotImplementedError(f"dropin: GPT-2 activation {act} is off the HIP hot path (erf gelu only)")
            if getattr(c, "scale_attn_by_inverse_layer_idx", False) or getattr(c, "reorder_and_upcast_attn", False) \
                    or getattr(c, "add_cross_attention", False) or not getattr(c, "scale_attn_weights", True):
                raise NotImplementedError("dropin: non-default GPT-2 attention options are off the HIP hot path")
            cfg = H.GPT2Config(n_embd=c.n_embd, n_head=c.n_head, n_layer=c.n_layer,
                               n_inner=c.n_inner if c.n_inner is not None else 4 * c.n_embd,
                               n_positions=c.n_positions, activation_function=act,
                               initializer_range=c.initializer_range, layer_norm_epsilon=c.layer_norm_epsilon,
                               resid_pdrop=c.resid_pdrop, embd_pdrop=c.embd_pdrop, attn_pdrop=c.attn_pdrop,
                               vocab_size=c.vocab_size)
            model = H.GPT2Model(cfg)
        elif kind == "BertModel":
            if getattr(c, "hidden_act", "gelu") != "gelu" or \
                    getattr(c, "position_embedding_type", "absolute") != "absolute" or getattr(c, "is_decoder", False):
                raise NotImplementedError("dropin: BERT hidden_act / position_embedding_type / is_decoder off the "
                                          "HIP hot path")
            cfg = H.BertConfig(hidden_size=c.hidden_size, num_hidden_layers=c.num_hidden_layers,
                               num_attention_heads=c.num_attention_heads, intermediate_size=c.intermediate_size,
                               max_position_embeddings=c.max_position_embeddings,
                               type_vocab_size=c.type_vocab_size, hidden_act="gelu",
                               initializer_range=c.initializer_range, layer_norm_eps=c.layer_norm_eps,
                               hidden_dropout_prob=c.hidden_dropout_prob,
                               attention_probs_dropout_prob=c.attention_probs_dropout_prob,
                               vocab_size=c.vocab_size)
            model = H.BertModel(cfg)
        else:
            raise NotImplementedError(f"dropin: transformer body {kind} is off the HIP hot path "
                                      "(XLNet / GPT-2 / BERT)")
        sh = X.TransformerBlock(model, masking=None)
    # HF registers non-parameter buffers (GPT-2 causal `bias`, BERT `position_ids`) the mirror does not have
    tie_by_name(sh.transformer, body, f"transformer body {kind}")
    return sh


# ------------------------------------------------------------------------------------------ task
class _EmbeddingsView:
    """what prediction_task.NextItemPredictionTask reads from `self.embeddings`"""

    def __init__(self, ref_cat):
        self._ref = ref_cat

    @property
    def item_seq(self):
        return self._ref.item_seq

    @property
    def item_embedding_table(self):
        return self._ref.item_embedding_table


def shadow_task(ref):
    """HIP-side NextItemPredictionTask sharing parameters / sampler buffers with the BUILT reference task."""
    if getattr(ref, "pre", None) is None or getattr(ref, "masking", None) is None:
        raise ValueError("dropin: the NextItemPredictionTask is not built yet (Head.build / to_torch_model first)")
    loss = ref.loss
    if not isinstance(loss, nn.CrossEntropyLoss) or getattr(loss, "reduction", "mean") != "mean" or \
            getattr(loss, "weight", None) is not None or getattr(loss, "ignore_index", -100) != -100:
        raise NotImplementedError("dropin: the HIP head fuses torch.nn.CrossEntropyLoss(mean) (optionally "
                                  "label-smoothed); other losses are off the hot path")
    pm = ref.pre.module
    # the reference task's torchmetrics objects, recognised by class name (ranking_metric.coerce), one by one: the
    # rank-based ones run on the fused evaluation head with the cut-offs the user configured; the others stay the
    # reference's business (its calculate_metrics still runs them).  metrics=[] stays "no metrics"; only a task
    # without the attribute gets the default set.
    ref_metrics = getattr(ref, "metrics", None)
    if ref_metrics is None:
        metrics = None
    else:
        metrics = []
        for m in ref_metrics:
            try:
                metrics.append(P.coerce_metric(m))
            except NotImplementedError:
                pass
    with _meta():
        sh = P.NextItemPredictionTask(loss=loss, metrics=metrics, task_name=ref.task_name, weight_tying=ref.weight_tying,
                                      softmax_temperature=ref.softmax_temperature, padding_idx=ref.padding_idx,
                                      target_dim=ref.target_dim, sampled_softmax=ref.sampled_softmax,
                                      max_n_samples=ref.max_n_samples)
        W = pm.item_embedding_table.weight if ref.weight_tying else pm.output_layer
        in_dim = int(W.shape[1])
        table = None
        if ref.weight_tying:
            table = F.EmbeddingTable(int(W.shape[0]), in_dim)
            table._parameters["weight"] = pm.item_embedding_table.weight
            sh.item_embedding_table = table
        tb = getattr(ref, "task_block", None)
        if tb is not None:
            try:
                dense = list(list(tb)[0])
                lin = dense[0]
                ok = len(list(tb)) == 1 and len(dense) == 1 and lin.weight.ndim == 2 and lin.bias is not None
            except (TypeError, AttributeError, IndexError):
                ok = False
            if not ok:
                raise NotImplementedError("dropin: only the automatic tied-weights task_block MLPBlock([item_dim], "
                                          "activation=None) is on the HIP hot path")
            sh.task_block = nn.Sequential(nn.Sequential(F._Linear(int(lin.weight.shape[1]), int(lin.weight.shape[0]))))
        sh.pre = P._Pre(P._NextItemPredictionModule(
            in_dim, ref.target_dim, ref.weight_tying, table, ref.softmax_temperature, ref.sampled_softmax,
            ref.max_n_samples, ref.padding_idx + 1))
    if ref.sampled_softmax:
        # the mirror derives the sampler's buffers from (max_id, min_id) on construction; on the meta
        # device they hold no data: share the reference's buffers (also what a loaded checkpoint restored)
        sm, rm = sh.pre.module.sampler, pm.sampler
        sm._buffers["dist"], sm._buffers["unique_sampling_dist"] = rm.dist, rm.unique_sampling_dist
    if tb is not None:
        tie_by_name(sh.task_block, ref.task_block, "task_block")
    if not ref.weight_tying:
        sh.pre.module._parameters["output_layer"] = pm.output_layer
    sh.embeddings = _EmbeddingsView(ref.embeddings)
    return sh


# ------------------------------------------------------------------------------------------ the subclasses
def _shadow_of(mod, builder):
    sh = mod.__dict__.get(_SHADOW)
    if sh is None:
        sh = builder(mod)
        object.__setattr__(mod, _SHADOW, sh)      # plain attribute: NOT a registered sub-module
    return sh


def drop_shadow(mod):
    """forget the cached shadow (after structural edits of the reference module)"""
    mod.__dict__.pop(_SHADOW, None)


_DP = {"acknowledged": False}
_DP_ATTR = "_t4r_hip_dp_ok"


def allow_data_parallel(flag=True):
    """Process-wide form of `enable_data_parallel`: the caller exchanges the `.grad` buffers of EVERY drop-in model of
    this process itself after backward (distributed.GradReducer / SparseRowExchange, or `sync_gradients`).  Lifts the
    world_size > 1 guard of the drop-in forward until `allow_data_parallel(False)`; must precede the first training forward."""
    _DP["acknowledged"] = bool(flag)


def enable_data_parallel(model, flag=True):
    """Per-model acknowledgement, to be given BEFORE the first training forward on world_size > 1: the caller promises to
    exchange this model's `.grad` buffers itself between backward() and optimizer.step() -- `sync_gradients(model)` (one flat
    averaged all-reduce) or distributed.GradReducer / SparseRowExchange.  The recipe is

        dropin.convert_model(model, data_parallel=True)      # or dropin.enable_data_parallel(model)
        loss = model(batch, training=True)["loss"]; loss.backward(); dropin.sync_gradients(model); optimizer.step()

    Marks every drop-in module inside `model` (a plain attribute, not in state_dict); returns the model."""
    for m in model.modules():
        if getattr(m, "_t4r_hip", False):
            object.__setattr__(m, _DP_ATTR, bool(flag))
    return model


def sync_gradients(model, group=None):
    """Data-parallel gradient exchange for a drop-in model: ONE flat all-reduce over every parameter's
    `.grad` (missing gradients count as zero), divided by the world size -- what torch DDP computes for the
    reference.  Call between `loss.backward()` and `optimizer.step()` on every rank.
    Works on any backend / device (gloo on CPU in tests, RCCL on the GPUs)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return model
    world = dist.get_world_size(group)
    if world == 1:
        return model
    params = [p for p in model.parameters() if p.requires_grad]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.mul_(1.0 / world)
    o = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[o:o + n].view_as(p.grad))
        o += n
    return model


def _check_data_parallel(mod, training):
    """raises when a training forward runs on > 1 ranks and nobody has taken charge of the gradients"""
    if not (training or mod.training) or _DP["acknowledged"] or mod.__dict__.get(_DP_ATTR, False):
        return
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        raise RuntimeError(
            "dropin: training on torch.distributed world_size > 1 with the FAST gradient path: the HIP backward writes parameter "
            "gradients into .grad without autograd hooks, so torch DDP (HF Trainer's wrapper) would see none of them.  Either "
            "(a) transformers4rec_amd.dropin.enable_ddp(model) -- or convert_model(model, ddp=True) -- BEFORE wrapping the model "
            "in torch.nn.parallel.DistributedDataParallel: the parameter gradients then travel through autograd "
            "(masking.GradCarrier; same kernels, same numbers) and DDP / HF Trainer work as for the reference; or "
            "(b) keep the fast path and take charge of the exchange yourself: dropin.enable_data_parallel(model) (or "
            "convert_model(model, data_parallel=True)), then dropin.sync_gradients(model) between backward() and "
            "optimizer.step(), or distributed.GradReducer / SparseRowExchange.")


def enable_ddp(model):
    """Autograd-visible parameter gradients for every drop-in module inside `model` (masking.enable_autograd_gradients) and the
    data-parallel acknowledgement: afterwards `torch.nn.parallel.DistributedDataParallel(model)` -- what HF Trainer wraps the
    reference's model in, transformers4rec/torch/trainer.py:131-161, docs/source/multi_gpu_train.md:27-40 -- averages the
    gradients of the HIP path through its own bucket hooks.  Call after convert_model / install, before wrapping."""
    from .masking import enable_autograd_gradients

    enable_autograd_gradients(model)
    return enable_data_parallel(model)


class _HipFeaturesMixin:
    _t4r_hip = True

    def hip_shadow(self):
        return _shadow_of(self, shadow_features)

    def forward(self, inputs, training=False, testing=False, **kwargs):
        _check_data_parallel(self, training)
        sh = self.hip_shadow()
        if sh.training != self.training:
            sh.train(self.training)
        out = sh(inputs, training=training, testing=testing)
        self.to_merge["categorical_module"].item_seq = sh.categorical_module.item_seq   # embedding.py:242-245
        m = self.masking
        if m is not None:      # what MaskSequence.forward leaves behind (masking.py:148-152)
            m.mask_schema, m.masked_targets = sh.masking.mask_schema, sh.masking.masked_targets
            object.__setattr__(m, "_t4r_hip_masking", sh.masking)
        return out


class _HipBlockMixin:
    _t4r_hip = True

    def hip_shadow(self):
        return _shadow_of(self, shadow_block)

    def forward(self, inputs_embeds, **kwargs):
        sh = self.hip_shadow()
        if sh.transformer.training != self.transformer.training:
            sh.transformer.train(self.transformer.training)
        return sh(inputs_embeds)


class _HipTaskMixin:
    _t4r_hip = True

    def hip_shadow(self):
        return _shadow_of(self, shadow_task)

    def forward(self, inputs, targets=None, training=False, testing=False, top_k=None, **kwargs):
        sh = self.hip_shadow()
        hm = self.masking.__dict__.get("_t4r_hip_masking") if self.masking is not None else None
        if (training or testing) and (hm is None or hm.masked_targets is not self.masking.masked_targets):
            raise RuntimeError("dropin: the HIP NextItemPredictionTask reads the label compaction of the HIP "
                               "TabularSequenceFeatures; install / convert both modules")
        sh.masking = hm if hm is not None else shadow_masking(self.masking)
        if sh.training != self.training:
            sh.train(self.training)
        return sh(inputs, targets=targets, training=training, testing=testing, top_k=top_k)


def make_dropin(tr):
    """-> (HipTabularSequenceFeatures, HipTransformerBlock, HipNextItemPredictionTask) subclassing the
    classes of the namespace `tr` (transformers4rec.torch)."""
    feats = type("HipTabularSequenceFeatures", (_HipFeaturesMixin, tr.TabularSequenceFeatures), {})
    block = type("HipTransformerBlock", (_HipBlockMixin, tr.TransformerBlock), {})
    task = type("HipNextItemPredictionTask", (_HipTaskMixin, tr.NextItemPredictionTask), {})
    for c in (feats, block, task):
        c.__module__ = __name__
    return feats, block, task


_INSTALLED = {}


def install(tr=None):
    """Replaces tr.TabularSequenceFeatures / TransformerBlock / NextItemPredictionTask by the HIP
    subclasses, so that the reference's own builders (`from_schema`, `XLNetConfig.to_torch_model`, which
    looks `TransformerBlock` up in this namespace, config/transformer.py:121-123) produce them.
    Returns the three classes; `uninstall(tr)` restores the originals."""
    if tr is None:
        import transformers4rec.torch as tr
    if id(tr) in _INSTALLED:
        return _INSTALLED[id(tr)][1]
    orig = (tr.TabularSequenceFeatures, tr.TransformerBlock, tr.NextItemPredictionTask)
    hip = make_dropin(tr)
    tr.TabularSequenceFeatures, tr.TransformerBlock, tr.NextItemPredictionTask = hip
    _INSTALLED[id(tr)] = (orig, hip, tr)     # holds the namespace: its id cannot be reused while installed
    return hip


def uninstall(tr=None):
    if tr is None:
        import transformers4rec.torch as tr
    rec = _INSTALLED.pop(id(tr), None)
    if rec is not None:
        tr.TabularSequenceFeatures, tr.TransformerBlock, tr.NextItemPredictionTask = rec[0]


def convert_model(model, tr=None, data_parallel=False):
    """Swaps, in place, the class of every reference TabularSequenceFeatures / TransformerBlock /
    NextItemPredictionTask inside `model` for its HIP subclass (no state is touched: the subclasses
    add methods only).  data_parallel=True also gives `enable_data_parallel(model)`'s acknowledgement.
    Returns the model."""
    if tr is None:
        import transformers4rec.torch as tr
    if id(tr) in _INSTALLED:
        orig, hip = _INSTALLED[id(tr)][:2]
    else:
        orig = (tr.TabularSequenceFeatures, tr.TransformerBlock, tr.NextItemPredictionTask)
        # keyed by id(), so the record keeps the namespace object ALIVE: a collected namespace's id can be handed to a
        # new object, which would then be converted with the old namespace's classes
        rec = _INSTALLED_CONVERT.get(id(tr))
        if rec is None or rec[0] is not tr or rec[1] != orig:
            rec = _INSTALLED_CONVERT[id(tr)] = (tr, orig, make_dropin(tr))
        _, orig, hip = rec
    for m in model.modules():
        if getattr(m, "_t4r_hip", False):
            continue
        for o, h in zip(orig, hip):
            if type(m) is o:
                m.__class__ = h
    if data_parallel:
        enable_data_parallel(model)
    return model


_INSTALLED_CONVERT = {}
