"""Masking modules: host-side mirror of transformers4rec/torch/masking.py for the hot path
(MaskSequence :61-243, CausalLanguageModeling :247-337, MaskedLanguageModeling :342-498).
Same constructor arguments, attributes (`mask_schema`, `masked_targets`, `padding_idx`,
`masked_item_embedding`) and state_dict names; the integer work runs in
csrc/masking.hip, the embedding replacement in csrc/embedding.hip.
PLM / RTD are out of scope.
"""
from collections import namedtuple
from typing import Optional

import torch
from torch import nn

from . import ops
from .rng import SeedMixin

MaskingInfo = namedtuple("MaskingInfo", ["schema", "targets"])


def _grad_buf(p):
    """The buffer a HIP backward ACCUMULATES the gradient of parameter `p` into: `p.grad` itself
    (created zero-filled on first use).  Contract of this package: the custom autograd functions
    write parameter gradients straight into `.grad` (flat-bucket friendly: optim.FlatParams makes
    `.grad` a view of the bucket RCCL reduces) and return None for them, so `torch.autograd.grad`
    and per-parameter hooks do not see parameter gradients of the HIP modules.
    A frozen parameter (`requires_grad=False`) keeps `.grad` untouched: the kernels get a scratch
    buffer that is cached on the parameter and never read."""
    c = p.__dict__.get("_t4r_carrier")
    if c is not None:          # autograd-visible mode (GradCarrier below): the gradient travels through the graph to p.grad
        return c
    if not p.requires_grad:
        s = p.__dict__.get("_t4r_frozen_scratch")
        if s is None or s.shape != p.shape or s.device != p.device:
            s = torch.zeros_like(p)
            p.__dict__["_t4r_frozen_scratch"] = s
        return s
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


class GradCarrier(torch.autograd.Function):
    """Makes the parameter gradients of the HIP modules VISIBLE TO AUTOGRAD (the reference trains under
    HF Trainer -> torch DistributedDataParallel, transformers4rec/torch/trainer.py:131-161, whose bucket hooks hang on the
    parameters' AccumulateGrad nodes).

    The fast path writes every parameter gradient straight into `.grad` (`_grad_buf`) and returns None to autograd: one Adam
    launch and one all-reduce per flat bucket, but no hook ever fires.  This node is the other mode, switched on per model
    (`enable_autograd_gradients`): it takes the parameters as inputs at the very FRONT of the forward (its output, one
    float, is the anchor of the input block's node, the first node of the path), hands every kernel a zero-filled carrier
    buffer instead of `.grad` for the duration of the step, and -- being the last node the engine runs in the backward pass,
    after the head, every layer and the input block have written their gradients -- returns those buffers as the parameters'
    gradients.  Autograd then accumulates them into `.grad` itself: AccumulateGrad runs, hooks run, torch DDP / per-parameter
    hooks / torch.autograd.grad see what they see for the reference.  Same kernels, same numbers; the cost is one
    zero fill of the parameters' size per step and autograd's accumulation pass (the fast path stays the default)."""

    @staticmethod
    def forward(ctx, owners, *params):
        """owners: the list of Parameter OBJECTS (the carriers hang on their __dict__, which `_grad_buf` reads); params: the
        same tensors as graph inputs"""
        ctx.owners = owners
        for p in owners:
            p.__dict__["_t4r_carrier"] = torch.zeros_like(p)
        return torch.zeros(1, device=owners[0].device)

    @staticmethod
    def backward(ctx, _d):
        from .transformer import _join_weight_gradient_streams

        _join_weight_gradient_streams()          # the layers' weight gradients run on side streams: in before they are read
        return (None,) + tuple(p.__dict__.pop("_t4r_carrier", None) for p in ctx.owners)

    @staticmethod
    def release(params):
        """a forward whose backward never ran (evaluation under grad mode, an exception): drop the carriers"""
        for p in params:
            p.__dict__.pop("_t4r_carrier", None)


def hip_parameters(model):
    """the parameters (requires_grad, each once) of every hot-path module inside `model`: the mirror classes of this package
    and the drop-in subclasses of the reference's"""
    from .features import TabularSequenceFeatures
    from .prediction_task import NextItemPredictionTask
    from .transformer import TransformerBlock

    seen, out = set(), []
    for m in model.modules():
        if isinstance(m, (TabularSequenceFeatures, TransformerBlock, NextItemPredictionTask)) or getattr(m, "_t4r_hip", False):
            for p in m.parameters():
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p))
                    out.append(p)
    return out


def enable_autograd_gradients(model, flag=True):
    """Switches the hot-path modules inside `model` to autograd-visible parameter gradients (GradCarrier): afterwards
    torch.nn.parallel.DistributedDataParallel(model), parameter hooks and plain torch optimizers over model.parameters() work as
    they do for the reference.  Call again after adding / freezing parameters.  Returns the model."""
    from .features import TabularSequenceFeatures

    params = hip_parameters(model) if flag else None
    n = 0
    for m in model.modules():
        is_hip_features = getattr(m, "_t4r_hip", False) and any(c.__name__ == "_HipFeaturesMixin" for c in type(m).__mro__)
        tgt = m.hip_shadow() if is_hip_features else m
        if isinstance(tgt, TabularSequenceFeatures):
            object.__setattr__(tgt, "_t4r_carrier_params", params)
            n += 1
    if flag and n == 0:
        raise ValueError("enable_autograd_gradients: no TabularSequenceFeatures (the first node of the hot path) inside the model")
    return model


class _ApplyMaskFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, memb, mask, mode):
        out = x.detach().clone()
        ops.apply_mask_fwd_(out, mask, memb.detach(), mode)
        ctx.mode = mode
        ctx.memb = memb
        ctx.save_for_backward(mask)
        return out

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        return ops.apply_mask_bwd(dy, mask, _grad_buf(ctx.memb), ctx.mode), None, None, None


class MaskSequence(SeedMixin, nn.Module):
    """Base class (reference masking.py:61-243)."""
    _seed_salt = 1

    def __init__(self, hidden_size: int, padding_idx: int = 0,
                 eval_on_last_item_seq_only: bool = True, **kwargs):
        super().__init__()
        self.padding_idx = padding_idx
        self.hidden_size = hidden_size
        self.eval_on_last_item_seq_only = eval_on_last_item_seq_only
        self.mask_schema: Optional[torch.Tensor] = None
        self.masked_targets: Optional[torch.Tensor] = None
        # trainable vector that replaces masked interactions; N(0, 0.001) (masking.py:102-108)
        self.masked_item_embedding = nn.Parameter(torch.empty(hidden_size))
        nn.init.normal_(self.masked_item_embedding, mean=0, std=0.001)
        # device-RNG state for the training draws (Philox key = `seed`, default rng.default_seed():
        # torch.initial_seed() + rank; counter `_rng_offset`, see rng.get_rng_state) and the parity hook
        self._rng_offset = 0
        self._draws = None
        # label compaction cache for the prediction head (filled by compute_masked_targets)
        self._row_count = None
        self._compact = None
        self._n_host = None
        self.last_item_ids = None

    # ---- parity hook: replay recorded torch.bernoulli / torch.multinomial draws once
    def set_draws(self, bern=None, j1=None, j2=None):
        self._draws = (bern, j1, j2)

    def _mode(self, training, testing):
        raise NotImplementedError

    def apply_mode(self, training, testing):
        raise NotImplementedError

    def compute_masked_targets(self, item_ids, training=False, testing=False) -> MaskingInfo:
        assert item_ids.ndim == 2, "`item_ids` must have 2 dimensions."
        mode = self._mode(training, testing)
        bern = j1 = j2 = None
        if self._draws is not None and mode == ops.MLM_TRAIN:
            bern, j1, j2 = self._draws
            self._draws = None
        mask, labels, counts = ops.mask_targets(
            item_ids.contiguous(), mode, self.padding_idx, bern, j1, j2,
            getattr(self, "mlm_probability", 0.0), self.seed, self._rng_offset)
        if mode == ops.MLM_TRAIN:
            self._rng_offset += item_ids.numel()
        self.mask_schema, self.masked_targets = mask, labels
        self.last_item_ids = item_ids            # read by TransformerBlock(mask_padding=True)
        self._row_count, self._compact = counts, None
        self._n_host = None
        if mode not in (ops.MLM_INFER, ops.CLM_INFER):
            # the head needs the number of label rows ON THE HOST (it shapes the compacted operands).  Compact
            # now and start the 4-byte copy: it completes while the host is still enqueueing the transformer,
            # so n_labels() at the head does not drain the device queue (a mid-step `.item()` would)
            n, _, _ = self.compact_labels()
            host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            host.copy_(n, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._n_host = (host, ev)
        return MaskingInfo(mask, labels)

    def n_labels(self) -> int:
        """number of label rows of the last compute_masked_targets (train / eval modes), on the host"""
        if self._n_host is None:
            return int(self.compact_labels()[0].item())
        host, ev = self._n_host
        ev.synchronize()
        return int(host[0])

    def compact_labels(self):
        """(n_labels [1] i32, label_pos [B*L] i32, labels [B*L] i64) in remove_pad_3d order."""
        if self._compact is None:
            self._compact = ops.compact_labels(self.masked_targets, self._row_count, self.padding_idx)
        return self._compact

    def apply_mask_to_inputs(self, inputs, schema, training=False, testing=False):
        raise NotImplementedError

    def forward(self, inputs, item_ids, training=False, testing=False):
        self.compute_masked_targets(item_ids=item_ids, training=training, testing=testing)
        if self.mask_schema is None:
            raise ValueError("`mask_schema must be set.`")
        return self.apply_mask_to_inputs(inputs, self.mask_schema, training=training, testing=testing)

    def forward_output_size(self, input_size):
        return input_size

    def transformer_required_arguments(self):
        return {}

    def transformer_optional_arguments(self):
        return {}

    @property
    def transformer_arguments(self):
        return {**self.transformer_required_arguments(), **self.transformer_optional_arguments()}


class CausalLanguageModeling(MaskSequence):
    """reference masking.py:247-337"""

    def __init__(self, hidden_size, padding_idx=0, eval_on_last_item_seq_only=True,
                 train_on_last_item_seq_only=False, **kwargs):
        super().__init__(hidden_size, padding_idx, eval_on_last_item_seq_only)
        self.train_on_last_item_seq_only = train_on_last_item_seq_only

    def _mode(self, training, testing):
        if not training and not testing:
            return ops.CLM_INFER
        last = (self.eval_on_last_item_seq_only and not training) or (
            self.train_on_last_item_seq_only and training)
        return ops.CLM_LAST if last else ops.CLM_TRAIN

    def apply_mode(self, training, testing):
        return ops.MASK_CLM_INFER if (not training and not testing) else ops.MASK_CLM

    def apply_mask_to_inputs(self, inputs, mask_schema, training=False, testing=False):
        return _ApplyMaskFn.apply(inputs.contiguous(), self.masked_item_embedding, mask_schema,
                                  self.apply_mode(training, testing))


class MaskedLanguageModeling(MaskSequence):
    """reference masking.py:342-498"""

    def __init__(self, hidden_size, padding_idx=0, eval_on_last_item_seq_only=True,
                 mlm_probability=0.15, **kwargs):
        super().__init__(hidden_size, padding_idx, eval_on_last_item_seq_only)
        self.mlm_probability = mlm_probability

    def _mode(self, training, testing):
        if training:
            return ops.MLM_TRAIN
        if testing:
            return ops.MLM_EVAL_LAST if self.eval_on_last_item_seq_only else ops.MLM_EVAL_ALL
        return ops.MLM_INFER

    def apply_mode(self, training, testing):
        return ops.MASK_MLM

    def apply_mask_to_inputs(self, inputs, mask_schema, training=False, testing=False):
        if not testing and not training:
            # inference: extend with a [MASK] slot (masking.py:489-492)
            inputs = torch.cat([inputs, inputs[:, -1:, :]], dim=1)
        return _ApplyMaskFn.apply(inputs.contiguous(), self.masked_item_embedding, mask_schema,
                                  ops.MASK_MLM)


masking_registry = {
    "clm": CausalLanguageModeling, "causal": CausalLanguageModeling,
    "mlm": MaskedLanguageModeling, "masked": MaskedLanguageModeling,
}


def parse_masking(masking, hidden_size, **kwargs):
    """masking_registry.parse(masking)(hidden_size=..., **kwargs) (features/sequence.py:221-224)"""
    if masking is None or isinstance(masking, MaskSequence):
        return masking
    if masking not in masking_registry:
        raise KeyError(f"{masking} never registered with registry masking (supported: clm, causal, mlm, masked)")
    return masking_registry[masking](hidden_size=hidden_size, **kwargs)
