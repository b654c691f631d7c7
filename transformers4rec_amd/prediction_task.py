"""NextItemPredictionTask: host-side mirror of transformers4rec/torch/model/prediction_task.py
  NextItemPredictionTask   :306-479   (constructor, build, forward train/eval/inference)
  _NextItemPredictionTask  :589-696   (tied / untied projection, temperature, sampled softmax)
  LogUniformSampler        :702-861
with the same arguments, output dict {"loss","labels","predictions"}, attribute contract and
state_dict names.  Arithmetic: csrc/masking.hip (row compaction), gemm_f32.hip
(X @ W^T and its two backward contractions), head.hip (softmax-CE, sampled logits, top-k).
"""
import math
from typing import Optional

import os

import torch
from torch import nn

from . import ops
from ._lib import exp_env as _exp_env
from .features import _Linear
from .masking import MaskedLanguageModeling, _grad_buf
from .ranking_metric import coerce as coerce_metric, default_metrics
from .rng import SeedMixin


class LogUniformSampler(SeedMixin, nn.Module):
    """reference prediction_task.py:702-861 (buffers `dist`, `unique_sampling_dist`).

    The draw: the reference calls torch.multinomial(self.dist, n_tries, replacement=True) -- a renormalisation and
    a prefix sum over all V categories per call (0.5 ms per step at 1 M items).  On the GPU the same categorical law
    is sampled by its closed-form inverse CDF (csrc/head.hip: log_uniform_sample_kernel) from a Philox stream
    (`seed`, default rng.default_seed(); position `_step`); `device_sampler = False` restores torch.multinomial."""
    _seed_salt = 5

    def __init__(self, max_n_samples: int, max_id: int, min_id: int = 0, unique_sampling: bool = True,
                 n_samples_multiplier_before_unique: int = 2):
        super().__init__()
        if max_id <= 0:
            raise ValueError("max_id must be a positive integer.")
        if max_n_samples <= 0:
            raise ValueError("n_sample must be a positive integer.")
        self.max_id, self.min_id, self.unique_sampling, self.max_n_samples = max_id, min_id, unique_sampling, max_n_samples
        self.device_sampler = True
        self._step = 0
        self.n_sample = int(max_n_samples * n_samples_multiplier_before_unique) if unique_sampling else max_n_samples
        with torch.no_grad():
            log_indices = torch.arange(1.0, max_id - min_id + 2.0, 1.0).log_()
            dist = (log_indices[1:] - log_indices[:-1]) / log_indices[-1]
            if min_id > 0:
                dist = torch.cat([torch.zeros([min_id], dtype=dist.dtype), dist], dim=0)
            self.register_buffer("dist", dist)
            self.register_buffer("unique_sampling_dist", (-(-dist.double().log1p_() * self.n_sample).expm1_()).float())

    def sample(self, labels: torch.Tensor):
        if not torch.is_tensor(labels):
            raise TypeError("Labels must be a torch.Tensor.")
        if labels.dtype != torch.long:
            raise ValueError("Labels must be a tensor of dtype long.")
        if labels.size(0) == 0:
            raise ValueError("Labels must not be an empty tensor.")
        with torch.no_grad():
            # negatives shared by the batch: multinomial with replacement, unique (sorted), truncated
            if self.device_sampler and self.dist.is_cuda:
                self._step += 1
                tries = ops.log_uniform_sample(self.n_sample, self.min_id, self.max_id, self.seed,
                                               ops.dropout_ctr_hi(self._step, 0xFC, 0), self.dist.device)
            else:
                tries = torch.multinomial(self.dist, self.n_sample, replacement=True)
            neg = tries.unique()[: self.max_n_samples] if self.unique_sampling else tries
            return neg.to(labels.device)

    @property
    def correction_dist(self):
        return self.unique_sampling_dist if self.unique_sampling else self.dist


class _NextItemPredictionModule(nn.Module):
    """Parameter holder named like the reference's `pre.module` (_NextItemPredictionTask)."""

    def __init__(self, input_dim, target_dim, weight_tying, item_embedding_table, softmax_temperature,
                 sampled_softmax, max_n_samples, min_id):
        super().__init__()
        self.target_dim, self.weight_tying = target_dim, weight_tying
        self.item_embedding_table = item_embedding_table
        self.softmax_temperature = softmax_temperature
        self.sampled_softmax = sampled_softmax
        if not weight_tying:
            self.output_layer = nn.Parameter(torch.empty(target_dim, input_dim))
            nn.init.kaiming_uniform_(self.output_layer, a=math.sqrt(5))
        if sampled_softmax:
            self.sampler = LogUniformSampler(max_n_samples=max_n_samples, max_id=target_dim, min_id=min_id,
                                             unique_sampling=True)

    @property
    def output_weights(self):
        return self.item_embedding_table.weight if self.weight_tying else self.output_layer


class _Pre(nn.Module):
    def __init__(self, module):
        super().__init__()
        self.module = module


# The head's d W runs on the caller's stream, right after d X.  (Rounds 1-2 ran it on a side stream under the body's
# backward; since round 3 the body's backward starts with token-tile kernels that take a whole CU each, the two only
# serialise -- 3.90 -> 3.85 ms per step without the side stream -- and a training step must not drive more than four
# streams, DESIGN.md section 6: the switch and its plumbing were removed in round 6.)
# sampled softmax: weight gradient as (ids, rows) + deterministic sorted scatter instead of row atomics
_SAMPLED_ROWS = _exp_env("T4R_SAMPLED_ROWS", "1") == "1"


_HEAD_SPLIT = _exp_env("T4R_HEAD_SPLIT", "1") != "0"
_HEAD_RECOMPUTE = _exp_env("T4R_HEAD_RECOMPUTE", "1") != "0"


def _head_split_ok(xp, W, N, V):
    """the materialised head for d_model <= 128 (csrc/head_split.hip): fp32-accurate precision modes only, shapes the
    general GEMM would also send to the split form (>= 2 GFLOP), 16-byte loadable operands"""
    D = W.shape[1]
    return (_HEAD_SPLIT and N > 0 and ops.get_precision() in ("auto", "fp32_bf16x3") and ops.head_split_supported(D)
            and 2.0 * N * V * D >= 2e9 and W.stride(1) == 1 and W.stride(0) % 4 == 0 and W.data_ptr() % 16 == 0
            and xp.is_contiguous() and xp.data_ptr() % 16 == 0)


class _NextItemHeadFn(torch.autograd.Function):
    """rows at label positions -> [task Linear] -> logits (full or sampled) -> mean CE."""

    @staticmethod
    def forward(ctx, x, anchor, task, pos, labels, N, neg):
        B, L, D = x.shape
        mod = task.pre.module
        W = mod.output_weights
        T = float(mod.softmax_temperature) if mod.softmax_temperature else 1.0
        x2 = x.contiguous().view(B * L, D)
        xr = ops.gather_rows(x2, pos, N)
        lin = task.task_block[0][0] if task.task_block is not None else None
        xp = xr
        if lin is not None:
            xp = ops.gemm(xr, lin.weight.detach(), False, True, bias=lin.bias.detach(), epilogue=ops.EPI_BIAS)
        V = W.shape[0]
        smooth = float(getattr(task.loss, "label_smoothing", 0.0) or 0.0)
        mode = task.resolve_head_mode(N, V) if neg is None else "materialize"
        ctx.recompute = (mode == "recompute" and any(ctx.needs_input_grad) and _head_split_ok(xp, W, N, V)
                         and ops.head_split_recompute_supported(W.shape[1]))
        if mode == "recompute" and not ctx.recompute:
            # the recomputing kernels cannot take this call (precision mode, T4R_HEAD_SPLIT=0, a misaligned table, a
            # training=True call under no_grad): fall back by SIZE -- the chunked head when the [N, V] scores do not
            # fit, never a silent multi-GB allocation
            mode = task.size_head_mode(N, V)
        ctx.fused = mode == "fused"
        if ctx.recompute:
            # no [N, V] tensor at all: statistics forward, score tiles recomputed by the two backward products
            # (csrc/head_split.hip: head_dw_rc / head_dx_rc kernels); `predictions` is computed if somebody reads it
            hws = ops.head_split_prepare(xp, V)
            loss, _rows, lse = ops.head_split_ce(hws, xp, W.detach(), labels, alpha=1.0 / T, label_smoothing=smooth)
            ctx.task, ctx.neg, ctx.meta = task, None, (B, L, D, N, V, T, V, smooth)
            ctx.hws = hws
            ctx.save_for_backward(pos, labels, labels, xr, xp, lse, lse)
            ctx.set_materialize_grads(False)
            return loss, None
        if ctx.fused:
            # non-materialising head: the vocabulary streams through one cache-sized [N, chunk] buffer,
            # online softmax statistics forward, chunk recomputation backward (csrc/head.hip)
            loss, _rows, lse = ops.linear_softmax_ce_fwd(xp, W.detach(), labels, 1.0 / T, smooth)
            ctx.task, ctx.neg, ctx.meta = task, None, (B, L, D, N, V, T, V, smooth)
            ctx.save_for_backward(pos, labels, labels, xr, xp, lse, lse)
            ctx.set_materialize_grads(False)
            return loss, None
        hws = None
        dx_unit = None
        if neg is None:
            if _head_split_ok(xp, W, N, V):
                # d_model <= 128: operands cut once, W-stationary logits (csrc/head_split.hip)
                # (and the softmax statistics reduced inside the product: no second pass over [N, V])
                hws = ops.head_split_prepare(xp, V)
                if ctx.needs_input_grad[0] and ops.head_split_fdx_supported(W.shape[1]) and W.stride(0) % 4 == 0:
                    # one pass (round 5): the scores come off the matrix cores once, are stored, and feed the d X product
                    # from registers -- the backward keeps only d W's read of the logits
                    logits, loss, _rows, lse, dx_unit = ops.head_split_logits_ce_dx(hws, xp, W.detach(), labels, alpha=1.0 / T,
                                                                                    label_smoothing=smooth, ldc=ops.pad_ld(V),
                                                                                    w_amax=ops.w_amax_of(W))
                else:
                    logits, loss, _rows, lse = ops.head_split_logits_ce(hws, xp, W.detach(), labels, alpha=1.0 / T,
                                                                        label_smoothing=smooth, ldc=ops.pad_ld(V))
            else:
                logits = ops.gemm(xp, W.detach(), False, True, alpha=1.0 / T, ldc=ops.pad_ld(V))
            tgt, width = labels, V
        else:
            logits = ops.sampled_logits_fwd(xp, labels, W.detach(), neg, mod.sampler.correction_dist, T)
            tgt, width = torch.zeros_like(labels), logits.shape[1]
        if hws is None:
            loss, _rows, lse = ops.softmax_ce_fwd(logits, tgt, width, smooth)
        ctx.task, ctx.neg, ctx.meta = task, neg, (B, L, D, N, V, T, width, smooth)
        ctx.hws = hws
        ctx.dx_unit = dx_unit
        ctx.save_for_backward(pos, labels, tgt, xr, xp, logits, lse)
        ctx.mark_non_differentiable(logits)
        # without this autograd hands backward() a zero-filled [N_m, V] gradient for `logits`
        # (a 1.1 GB fill, ~150 us per step at the 100k-item configuration)
        ctx.set_materialize_grads(False)
        return loss, logits

    @staticmethod
    def backward(ctx, dloss, _dlogits_unused):
        pos, labels, tgt, xr, xp, logits, lse = ctx.saved_tensors
        task = ctx.task
        B, L, D, N, V, T, width, smooth = ctx.meta
        mod = task.pre.module
        W = mod.output_weights
        if dloss is None:
            return (None,) * 7
        dense_dx = False
        if getattr(ctx, "recompute", False):
            g = dloss.contiguous()
            hws = ctx.hws
            dxp = ops.head_split_dx_rc(hws, xp, W.detach(), lse, labels, g, alpha=1.0 / T, label_smoothing=smooth)
            if W.requires_grad:
                ops.head_split_dw_rc(hws, W.detach(), lse, labels, g, _grad_buf(W), alpha=1.0 / T, label_smoothing=smooth,
                                     accumulate=True)
        elif ctx.fused:
            dxp = ops.linear_softmax_ce_bwd(xp, W.detach(), labels, lse, dloss.contiguous(),
                                            dW=_grad_buf(W) if W.requires_grad else None, alpha=1.0 / T,
                                            label_smoothing=smooth)
        elif ctx.neg is None:
            # CrossEntropyLoss backward is fused into the A operand of both contractions:
            # the [N_m, V] gradient is never materialised
            g = dloss.contiguous()
            hws = ctx.hws
            Dh = W.shape[1]
            if getattr(ctx, "dx_unit", None) is not None:
                # formed in the forward for an upstream gradient of 1.  Without a task block the scaling by g, the zero fill of
                # the [B L, D] gradient and the row scatter are ONE launch (csrc/masking.hip: scatter_rows_dense_kernel)
                dense_dx = task.task_block is None
                dxp = ctx.dx_unit if dense_dx else ctx.dx_unit * g
                ctx.dx_unit = None
            elif hws is not None:
                dxp = ops.head_split_dx(hws, logits, lse, tgt, g, V, W.detach(), alpha=1.0 / T, label_smoothing=smooth)
            else:
                dxp = ops.gemm_softmax_grad(logits, lse, tgt, g, V, W.detach(), False, alpha=1.0 / T,
                                            label_smoothing=smooth, splitk=-1)

            def d_w(gw):
                if hws is not None:
                    ops.head_split_dw(hws, logits, lse, tgt, g, V, Dh, gw, alpha=1.0 / T, label_smoothing=smooth,
                                      accumulate=True)
                else:
                    ops.gemm_softmax_grad(logits, lse, tgt, g, V, xp, True, alpha=1.0 / T, label_smoothing=smooth,
                                          out=gw, accumulate=True)
            if W.requires_grad:
                d_w(_grad_buf(W))
        else:
            dl = ops.softmax_ce_bwd(logits, tgt, lse, dloss.contiguous(), width, smooth)
            sink = getattr(W, "_t4r_sparse_sink", None)
            if sink is not None or (_SAMPLED_ROWS and W.requires_grad):
                # row-sparse weight gradient: (N + S) rows instead of atomics into a dense [V, D] buffer;
                # summed deterministically here, or exchanged between data-parallel ranks by the sink
                dxp, ids, rows = ops.sampled_logits_bwd_rows(dl, xp, labels, W.detach(), ctx.neg, T)
                if sink is not None:
                    sink.add_rows(W, ids, rows, padding_idx=-1)
                else:
                    ops.scatter_rows_sorted(_grad_buf(W), ids, rows)
            else:
                dxp = ops.sampled_logits_bwd(dl, xp, labels, W.detach(), ctx.neg, _grad_buf(W), T)
        lin = task.task_block[0][0] if task.task_block is not None else None
        dxr = dxp
        if lin is not None:
            ops.gemm(dxp, xr, True, False, splitk=-1, accumulate=True, out=_grad_buf(lin.weight))
            ops.colsum_(dxp, _grad_buf(lin.bias))
            dxr = ops.gemm(dxp, lin.weight.detach(), False, False)
        if dense_dx:
            dx = ops.scatter_rows_dense(dxr, pos, N, g, B * L)
        else:
            dx = torch.zeros((B * L, D), device=dxr.device, dtype=torch.float32)
            ops.scatter_rows_add_(dxr, pos, dx)
        return dx.view(B, L, D), None, None, None, None, None, None


class LazyPredictions:
    """`predictions` of the non-materialising head: the [N, V] logits are computed (one GEMM) the first
    time anything looks at them -- `.materialize()`, any tensor attribute / method, indexing, or any
    torch function taking this object -- and cached.  Shape / dtype / device are known without computing."""

    def __init__(self, compute, shape, device):
        self._compute, self._value = compute, None
        self.shape, self.device, self.dtype = torch.Size(shape), device, torch.float32

    def materialize(self):
        if self._value is None:
            self._value = self._compute()
            self._compute = None
        return self._value

    @property
    def is_materialized(self):
        return self._value is not None

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return len(self.shape)

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, idx):
        return self.materialize()[idx]

    def __getattr__(self, name):         # only reached for names not defined above: tensor methods
        if name.startswith("__") or name in ("_compute", "_value"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        un = lambda a: a.materialize() if isinstance(a, LazyPredictions) else a
        args = tuple(un(a) for a in args)
        kwargs = {k: un(v) for k, v in (kwargs or {}).items()}
        return func(*args, **kwargs)

    def __repr__(self):
        state = "materialized" if self.is_materialized else "not computed"
        return f"LazyPredictions(shape={tuple(self.shape)}, {state})"


class NextItemPredictionTask(nn.Module):
    """Drop-in for tr.NextItemPredictionTask on the hot path."""

    def __init__(self, loss: nn.Module = None, metrics=None, task_block=None, task_name: str = "next-item",
                 weight_tying: bool = False, softmax_temperature: float = 1, padding_idx: int = 0,
                 target_dim: int = None, sampled_softmax: Optional[bool] = False,
                 max_n_samples: Optional[int] = 100, top_ks=(10, 20), head_mode: str = "auto"):
        super().__init__()
        if head_mode not in ("auto", "materialize", "fused", "recompute"):
            raise ValueError("head_mode must be 'auto', 'materialize', 'fused' or 'recompute'")
        # full-softmax training / evaluation head (beyond the reference's signature):
        #   "materialize": logits [N, V] in HBM, `predictions` is that tensor (what the reference returns)
        #   "fused"      : no [N, V] tensor; `predictions` is a LazyPredictions that computes them on first use
        #   "auto"       : materialise while the logits are small (<= T4R_HEAD_AUTO_GB, default 4 GB), else fused
        self.head_mode = head_mode
        loss = loss if loss is not None else nn.CrossEntropyLoss()
        if not isinstance(loss, nn.CrossEntropyLoss):
            raise NotImplementedError("the HIP head fuses torch.nn.CrossEntropyLoss (optionally label-smoothed)")
        if task_block is not None:
            raise NotImplementedError("custom task_block is off the hot path (auto projection is supported)")
        self.loss = loss
        self.task_name = task_name
        self.softmax_temperature = softmax_temperature
        self.weight_tying = weight_tying
        self.padding_idx = padding_idx
        self.target_dim = target_dim
        self.sampled_softmax = sampled_softmax
        self.max_n_samples = max_n_samples
        self.top_ks = tuple(top_ks)
        # metrics: rank-based descriptors (ranking_metric.py of this package), registry names, or the reference's own
        # metric objects (recognised by class name); default = the reference's DEFAULT_METRICS at `top_ks`
        self.metrics = tuple(default_metrics(self.top_ks) if metrics is None
                             else [coerce_metric(m, self.top_ks) for m in metrics])
        self.task_block = None
        self.item_embedding_table = None
        self.masking = None
        self.embeddings = None
        self.pre = None
        self._metric_acc = None

    def build(self, body=None, input_size=None, device=None, inputs=None, task_block=None, pre=None):
        """Called by the Head/Model (prediction_task.py:369-417)."""
        if input_size is None or len(input_size) != 3:
            raise ValueError(f"NextItemPredictionTask needs a 3-dim vector as input, found:{input_size}")
        if inputs is None:
            inputs = body.inputs
        if not getattr(inputs, "item_id", None):
            raise ValueError("For Item Prediction task a categorical_module including an item_id column is required.")
        self.embeddings = inputs.categorical_module
        n_items = self.embeddings.item_embedding_table.num_embeddings
        if not self.target_dim:
            self.target_dim = n_items
        if self.target_dim < n_items:
            # the labels ARE item ids (masking.masked_targets): the head kernels index logits / W rows with
            # them without a range check of their own (the ids themselves are range-checked by the embedding
            # gather, TabularSequenceFeatures.check_ids()).  torch's CrossEntropyLoss would raise at run time.
            raise ValueError(f"target_dim ({self.target_dim}) must cover the item-id cardinality ({n_items}): "
                             "labels are item ids")
        item_table = None
        in_dim = input_size[-1]
        if self.weight_tying:
            item_table = self.embeddings.item_embedding_table
            self.item_embedding_table = item_table
            item_dim = item_table.weight.shape[1]
            if in_dim != item_dim:
                # MLPBlock([item_dim], activation=None)  (prediction_task.py:390-397)
                self.task_block = nn.Sequential(nn.Sequential(_Linear(in_dim, item_dim)))
                in_dim = item_dim
        self.masking = inputs.masking
        if not self.masking:
            raise ValueError("The input block should contain a masking schema for training and evaluation")
        self.padding_idx = self.masking.padding_idx
        self.pre = _Pre(_NextItemPredictionModule(
            in_dim, self.target_dim, self.weight_tying, item_table, self.softmax_temperature,
            self.sampled_softmax, self.max_n_samples, self.padding_idx + 1))
        if device is not None:
            self.to(device)
        return self

    @staticmethod
    def size_head_mode(N, V):
        """'materialize' when the [N, V] fp32 scores fit T4R_HEAD_AUTO_GB (default 4 GiB), else the chunked 'fused' head"""
        limit = float(os.environ.get("T4R_HEAD_AUTO_GB", "4")) * (1 << 30)
        return "materialize" if 4.0 * N * ops.pad_ld(V) <= limit else "fused"

    def resolve_head_mode(self, N, V):
        mode = os.environ.get("T4R_HEAD_MODE") or self.head_mode
        if mode == "auto":
            mode = self.size_head_mode(N, V)
            # Scores too large to keep: at a head_split.hip width the RECOMPUTING head (round 4: statistics forward, score
            # tiles recomputed by the two backward products, csrc/head_split.hip) replaces the chunked general path.
            # Where the scores do fit it is NOT the default: measured at BASELINE configs[1] (d_model 128) the two extra
            # products cost more than the 3.3 GB of logits traffic they save (step 3.13 vs 2.89 ms: docs/DESIGN_rounds_1_to_4.md, round 4);
            # head_mode="recompute" / T4R_HEAD_MODE=recompute selects it anyway (training calls only: metrics read the scores)
            if mode == "fused" and getattr(self, "_training_call", False) and _HEAD_RECOMPUTE:
                D = self.pre.module.output_weights.shape[1]
                # its workspace (per-tile statistics + two table images) is not cache-sized like the chunked head's one
                # chunk: ~30 GB at V = 10 M, N = 15 k, D = 128 -- above T4R_HEAD_WS_GB (default 16) the chunked head stays
                ws_limit = float(os.environ.get("T4R_HEAD_WS_GB", "16")) * (1 << 30)
                if (D <= 128 and ops.head_split_recompute_supported(D)
                        and ops.head_split_ws_bytes(N, V, D) <= ws_limit):
                    mode = "recompute"
        if mode == "recompute" and not getattr(self, "_training_call", False):
            # evaluation calls read the scores: by size, as `auto` (materialize if they fit, else the chunked head)
            mode = self.size_head_mode(N, V)
        return mode

    def _lazy_predictions(self, x, pos, labels, N):
        mod = self.pre.module
        W = mod.output_weights
        T = float(mod.softmax_temperature) if mod.softmax_temperature else 1.0
        xd = x.detach()

        def compute():
            B, L, D = xd.shape
            xr = ops.gather_rows(xd.contiguous().view(B * L, D), pos, N)
            if self.task_block is not None:
                lin = self.task_block[0][0]
                xr = ops.gemm(xr, lin.weight.detach(), False, True, bias=lin.bias.detach(), epilogue=ops.EPI_BIAS)
            V = W.shape[0]
            return ops.gemm(xr, W.detach(), False, True, alpha=1.0 / T, ldc=ops.pad_ld(V))[:, :V]

        return LazyPredictions(compute, (N, W.shape[0]), x.device)

    # ------------------------------------------------------------------ forward
    def forward(self, inputs, targets=None, training=False, testing=False, top_k=None, **kwargs):
        if isinstance(inputs, (tuple, list)):
            inputs = inputs[0]
        x = inputs.float()
        mod = self.pre.module
        if training or testing:
            self._training_call = bool(training)
            n, pos, lab = self.masking.compact_labels()
            # N shapes the row-compacted operands; it was copied to the host right after the masking kernel
            # (masking.n_labels): by now it has long arrived, so this does not drain the queue
            N = self.masking.n_labels()
            labels = lab[:N]
            if N == 0:
                raise ValueError("no label positions in this batch")
            neg = mod.sampler.sample(labels) if (self.sampled_softmax and training) else None
            loss, logits = _NextItemHeadFn.apply(x, mod.output_weights, self, pos, labels, N, neg)
            if logits is None:          # non-materialising head: the logits exist only if somebody asks
                preds, y = self._lazy_predictions(x, pos, labels, N), labels
            elif neg is None:
                preds = logits[:, : mod.output_weights.shape[0]]
                y = labels
            else:
                preds, y = logits, torch.zeros_like(labels)
            return {"loss": loss, "labels": y, "predictions": preds}
        # inference: hidden state at the last item (prediction_task.py:452-470)
        item_seq = self.embeddings.item_seq
        B, Lg, D = x.shape
        pos = ops.last_positions(item_seq.contiguous(), Lg, isinstance(self.masking, MaskedLanguageModeling),
                                 self.padding_idx)
        xr = ops.gather_rows(x.contiguous().view(B * Lg, D), pos, B)
        if self.task_block is not None:
            lin = self.task_block[0][0]
            xr = ops.gemm(xr, lin.weight.detach(), False, True, bias=lin.bias.detach(), epilogue=ops.EPI_BIAS)
        W = mod.output_weights.detach()
        T = float(mod.softmax_temperature) if mod.softmax_temperature else 1.0
        V = W.shape[0]
        if not torch.is_grad_enabled():      # registered operators (torch_ops.py): dispatcher-visible inference head
            from . import torch_ops  # noqa: F401

            scores = torch.ops.t4r_hip.item_scores(xr, W, 1.0 / T)
            return scores if top_k is None else torch.ops.t4r_hip.topk(scores, top_k)
        scores = ops.gemm(xr, W, False, True, alpha=1.0 / T, ldc=ops.pad_ld(V))
        if top_k is None:
            return scores
        return ops.topk(scores, top_k, V)

    # ------------------------------------------------------------------ ranking metrics (N1)
    def _rank_metrics(self, ranks):
        """{"<metric>_<k>": per-row values} for the task's metrics, from 0-based target ranks, and their
        accumulation for compute_metrics(): (sum, count) per entry, kept ON THE DEVICE in fp64 -- no host
        synchronisation per batch."""
        out = {}
        for m in self.metrics:
            vals = m.from_ranks(ranks)
            for j, k in enumerate(m.top_ks):
                out[f"{m.name}_{k}"] = vals[:, j]
        names = self._metric_names()
        add = torch.stack([out[n].sum(dtype=torch.float64) for n in names]
                          + [torch.tensor(float(ranks.numel()), dtype=torch.float64, device=ranks.device)])
        self._metric_acc = add if self._metric_acc is None else self._metric_acc + add
        return out

    def _metric_names(self):
        return [f"{m.name}_{k}" for m in self.metrics for k in m.top_ks]

    def calculate_metrics(self, predictions, targets):
        """The task's ranking metrics with one relevant item per row (ranking_metric.py:52-59 labels_onehot=True)
        from a fused top-k of the scores -- no [N, V] one-hot.  A target outside the top max(k) gets rank
        max(k): every metric of the path is zero from there on."""
        if not self.metrics:
            return {}
        kmax = max(k for m in self.metrics for k in m.top_ks)
        _, idx = ops.topk(predictions, kmax, predictions.shape[1])
        hit = idx == targets.unsqueeze(-1)
        pos = torch.arange(kmax, device=idx.device, dtype=torch.int64)
        ranks = torch.where(hit, pos, torch.full_like(pos, kmax)).min(dim=-1).values
        return self._rank_metrics(ranks)

    def metrics_from_ranks(self, ranks):
        """The task's ranking metrics from 0-based target ranks (one relevant item per row)."""
        return self._rank_metrics(ranks.to(torch.int64))

    def evaluate_ranks(self, inputs):
        """Fused evaluation head: the label rows of `inputs` [B, L, D] (the masking's
        evaluation targets, e.g. the last item of every session) -> rank of the target item among
        all V scores, computed tile by tile inside the logits GEMM; the [N, V] score matrix, the
        top-k pass and the reference's [N, V] one-hot (ranking_metric.py:52-59) never exist.
        Returns {"labels", "ranks", "metrics"}; metrics are also accumulated for compute_metrics()."""
        x = (inputs[0] if isinstance(inputs, (tuple, list)) else inputs).float()
        mod = self.pre.module
        n, pos, lab = self.masking.compact_labels()
        N = self.masking.n_labels()
        if N == 0:
            raise ValueError("no label positions in this batch")
        labels = lab[:N]
        B, L, D = x.shape
        xr = ops.gather_rows(x.detach().contiguous().view(B * L, D), pos, N)
        if self.task_block is not None:
            lin = self.task_block[0][0]
            xr = ops.gemm(xr, lin.weight.detach(), False, True, bias=lin.bias.detach(), epilogue=ops.EPI_BIAS)
        T = float(mod.softmax_temperature) if mod.softmax_temperature else 1.0
        ranks = ops.rank_of_target(xr, mod.output_weights.detach(), labels, 1.0 / T)
        return {"labels": labels, "ranks": ranks, "metrics": self.metrics_from_ranks(ranks)}

    def compute_metrics(self, mode=None, group=None):
        """{"<task>/<metric>_<k>": mean over ALL rows seen since reset_metrics()}.  Under torch.distributed
        (world_size > 1) the (sum, count) state is all-reduced first -- the reference `cat`-synchronises the
        per-row values of its torchmetrics state and averages them (ranking_metric.py:50,64-66; gathered by HF
        Trainer's evaluation loop, torch/trainer.py:519-525): the same mean over every rank's rows.
        COLLECTIVE: every rank must call it (a rank that evaluated nothing contributes zeros)."""
        import torch.distributed as dist

        names = self._metric_names()
        acc = self._metric_acc
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        if distributed:
            if acc is None:
                dev = next((p.device for p in self.parameters()), None) or \
                    self.pre.module.output_weights.device
                if dist.get_backend(group) == "nccl" and dev.type != "cuda":
                    dev = torch.device("cuda", torch.cuda.current_device())
                acc = torch.zeros(len(names) + 1, dtype=torch.float64, device=dev)
            acc = acc.clone()
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
        if acc is None:
            return {}
        vals = acc.cpu().tolist()
        n = max(vals[-1], 1.0)
        return {f"{self.task_name}/{k}": v / n for k, v in zip(names, vals[:-1])}

    def reset_metrics(self):
        self._metric_acc = None

    def metric_name(self, name):
        return f"{self.task_name}/{name}"
