"""TransformerBlock + XLNet configuration: host-side mirror of
  transformers4rec/torch/block/transformer.py:76-206   (TransformerBlock)
  transformers4rec/config/transformer.py:423-482        (XLNetConfig.build)
and of the parameter layout of HF `XLNetModel` (third-party dependency of the reference;
modeling_xlnet.py:245-353, 979-1205) so that reference checkpoints load unchanged
(state_dict names `transformer.layer.<i>.rel_attn.{q,k,v,o,r,r_r_bias,r_s_bias,r_w_bias,
seg_embed,layer_norm.*}`, `...ff.{layer_norm,layer_1,layer_2}.*`, `transformer.mask_emb`,
`transformer.word_embedding.weight`).  The arithmetic is csrc/xlnet_layer.hip.

Reference semantics kept on purpose: NO padding attention mask; the
relative positional keys are computed once per layer, not per batch row.
"""
from dataclasses import dataclass, field
from typing import Optional

import os

import torch
from torch import nn

from . import ops
from .masking import MaskedLanguageModeling, CausalLanguageModeling, MaskSequence, _grad_buf
from .rng import SeedMixin


@dataclass
class XLNetConfig:
    """Fields XLNetConfig.build sets on the HF config (config/transformer.py:432-482)."""
    d_model: int = 128
    n_head: int = 4
    n_layer: int = 4
    d_inner: int = 512
    attn_type: str = "bi"
    ff_activation: str = "gelu"
    initializer_range: float = 0.01
    layer_norm_eps: float = 0.03
    dropout: float = 0.3
    pad_token_id: int = 0
    vocab_size: int = 1
    mem_len: int = 1
    total_seq_length: Optional[int] = None
    model_type: str = field(default="xlnet", repr=False)

    @classmethod
    def build(cls, d_model, n_head, n_layer, total_seq_length=None, attn_type="bi", hidden_act="gelu",
              initializer_range=0.01, layer_norm_eps=0.03, dropout=0.3, pad_token=0,
              log_attention_weights=False, mem_len=1, **kwargs):
        if attn_type != "bi":
            raise NotImplementedError("only attn_type='bi' (the reference default) is on the hot path")
        if hidden_act != "gelu":
            raise NotImplementedError("only hidden_act='gelu' (erf) is on the hot path")
        if d_model % n_head:
            raise ValueError(f"The hidden size ({d_model}) is not a multiple of the number of attention heads ({n_head}")
        return cls(d_model=d_model, n_head=n_head, n_layer=n_layer, d_inner=4 * d_model,
                   attn_type=attn_type, ff_activation=hidden_act, initializer_range=initializer_range,
                   layer_norm_eps=layer_norm_eps, dropout=dropout, pad_token_id=pad_token,
                   mem_len=mem_len, total_seq_length=total_seq_length)

    @property
    def hidden_size(self):
        return self.d_model

    @property
    def d_head(self):
        return self.d_model // self.n_head

    def to_huggingface_torch_model(self):
        return XLNetModel(self)

    def to_torch_model(self, input_features, *prediction_task, task_blocks=None, task_weights=None,
                       loss_reduction="mean", **kwargs):
        """config/transformer.py:71-131: SequentialBlock(inputs, TransformerBlock) -> Head -> Model."""
        from .model import Model

        if len(prediction_task) != 1:
            raise NotImplementedError("one NextItemPredictionTask per model on the hot path")
        block = TransformerBlock(self, masking=input_features.masking)
        return Model(input_features, block, prediction_task[0])


class _LN(nn.Module):
    def __init__(self, dim, eps):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _Lin(nn.Module):
    def __init__(self, fin, fout, std):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin).normal_(0, std))
        self.bias = nn.Parameter(torch.zeros(fout))


class XLNetRelativeAttentionParams(nn.Module):
    def __init__(self, cfg: XLNetConfig):
        super().__init__()
        D, n, dh, s = cfg.d_model, cfg.n_head, cfg.d_head, cfg.initializer_range
        mk = lambda *shape: nn.Parameter(torch.empty(*shape).normal_(0, s))
        self.q, self.k, self.v, self.o, self.r = mk(D, n, dh), mk(D, n, dh), mk(D, n, dh), mk(D, n, dh), mk(D, n, dh)
        self.r_r_bias, self.r_s_bias, self.r_w_bias = mk(n, dh), mk(n, dh), mk(n, dh)
        self.seg_embed = mk(2, n, dh)      # unused without token_type_ids (as in the reference run)
        self.layer_norm = _LN(D, cfg.layer_norm_eps)


class XLNetFeedForwardParams(nn.Module):
    def __init__(self, cfg: XLNetConfig):
        super().__init__()
        self.layer_norm = _LN(cfg.d_model, cfg.layer_norm_eps)
        self.layer_1 = _Lin(cfg.d_model, cfg.d_inner, cfg.initializer_range)
        self.layer_2 = _Lin(cfg.d_inner, cfg.d_model, cfg.initializer_range)


class XLNetLayer(nn.Module):
    def __init__(self, cfg: XLNetConfig):
        super().__init__()
        self.rel_attn = XLNetRelativeAttentionParams(cfg)
        self.ff = XLNetFeedForwardParams(cfg)

    def ordered_params(self):
        a, f = self.rel_attn, self.ff
        return [a.q, a.k, a.v, a.o, a.r, a.r_w_bias, a.r_r_bias, a.layer_norm.weight, a.layer_norm.bias,
                f.layer_1.weight, f.layer_1.bias, f.layer_2.weight, f.layer_2.bias, f.layer_norm.weight,
                f.layer_norm.bias]


def relative_positional_encoding(L, D):
    """HF modeling_xlnet.py:940-976 for attn_type='bi', bi_data=False, clamp_len=-1, klen==qlen:
    pos_seq = arange(L, -L, -1) ; [sin(pos*inv_freq) | cos(pos*inv_freq)]  -> [2L, D]"""
    freq_seq = torch.arange(0, D, 2.0, dtype=torch.int64).float()
    inv_freq = 1 / torch.pow(10000, (freq_seq / D))
    pos_seq = torch.arange(L, -L, -1.0, dtype=torch.int64).float()
    sinusoid = torch.einsum("i,d->id", pos_seq, inv_freq)
    return torch.cat([torch.sin(sinusoid), torch.cos(sinusoid)], dim=-1)


class _XLNetLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, anchor, layer, pos_emb, n_head, eps, drop, key_len=None, pos_emb_b=None, ws=None):
        """ws: this layer's workspace with its weight planes and k_r already in it (XLNetModel.forward: stack prologue)"""
        B, L, D = h.shape
        h2 = h.contiguous().view(B * L, D)
        params = [p.detach() for p in layer.ordered_params()]
        p, seed, offset, idx = drop
        out, ws = ops.xlnet_layer_fwd(h2, pos_emb, params, B, L, n_head, eps, ws=ws, drop_p=p, seed=seed,
                                      offset=offset, layer_idx=idx, key_len=key_len, pos_emb_b=pos_emb_b,
                                      stack_prepared=ws is not None)
        ctx.layer, ctx.pos_emb, ctx.cfg, ctx.drop, ctx.key_len = layer, pos_emb, (B, L, D, n_head, eps), drop, key_len
        ctx.pos_emb_b = pos_emb_b
        ctx.save_for_backward(h2, ws)
        return out.view(B, L, D)

    @staticmethod
    def backward(ctx, dout):
        h2, ws = ctx.saved_tensors
        B, L, D, n_head, eps = ctx.cfg
        p, seed, offset, idx = ctx.drop
        plist = ctx.layer.ordered_params()
        grads = [_grad_buf(q) for q in plist]
        # the weight-gradient streams of this layer are joined at the END of the backward pass, not here: their tail
        # (r, q|k|v, the reduction) runs under the next layer's backward instead of stalling it (csrc/xlnet_layer.hip:
        # deferred join).  The buffers they use stay referenced in _PENDING until the join.
        defer = _PENDING if _DEFER_JOIN else None
        dh = ops.xlnet_layer_bwd(h2, ctx.pos_emb, [q.detach() for q in plist], grads, ws,
                                 dout.contiguous().view(B * L, D), B, L, n_head, eps, drop_p=p, seed=seed,
                                 offset=offset, layer_idx=idx, key_len=ctx.key_len, pos_emb_b=ctx.pos_emb_b,
                                 defer_join=defer)
        if defer is not None:
            # one callback per deferred call (idempotent): no state that an aborted backward pass could leave behind
            torch.autograd.Variable._execution_engine.queue_callback(_join_weight_gradient_streams)
        return dh.view(B, L, D), None, None, None, None, None, None, None, None, None


from ._lib import exp_env as _exp_env  # noqa: E402

_DEFER_JOIN = _exp_env("T4R_XLNET_DEFER_JOIN", "1") != "0"       # a test flips the attribute; the product reads no switch
_FUSED_ON = _exp_env("T4R_XLNET_FUSED", "1") != "0"
_FUSE_FINAL = True       # module attribute (no switch): tools/ab_step.py flips it for a same-box A/B
_FUSE_INPUT = True
_GEN_POS = True          # the dropped positional rows are made by the stack prologue's projection kernel
_STACK_PROLOGUE = _exp_env("T4R_XLNET_STACK_PROLOGUE", "1") != "0" and _FUSED_ON
_PENDING: list = []          # buffers of deferred layer backwards (kept alive until the join)


def _join_weight_gradient_streams():
    """end-of-backward callback of the autograd engine: runs before backward() returns, on the caller's current stream.
    Also called defensively at the start of the next XLNetModel.forward and from FusedAdam.step / zero_grad: a backward
    pass that RAISED may never have run its callbacks, and code that catches the exception and then reads or zeroes
    .grad, or skips the step, must not race with weight-gradient GEMMs still in flight (idempotent: no-op when nothing
    is pending)."""
    if not _PENDING:
        return
    try:
        ops.xlnet_layer_bwd_join()
    finally:
        _PENDING.clear()


class _DropoutFn(torch.autograd.Function):
    """element-wise dropout with a recomputed Philox mask (model-level sites of HF XLNetModel:
    inputs_embeds :1116, final output :1177)"""

    @staticmethod
    def forward(ctx, x, p, seed, ctr_hi):
        ctx.cfg = (p, seed, ctr_hi)
        return ops.dropout(x.contiguous().view(-1), p, seed, ctr_hi).view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        p, seed, ctr_hi = ctx.cfg
        return ops.dropout(dy.contiguous().view(-1), p, seed, ctr_hi).view(dy.shape), None, None, None


class _WordEmbedding(nn.Module):
    def __init__(self, vocab, dim, std):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(vocab, dim).normal_(0, std))


class XLNetModel(SeedMixin, nn.Module):
    """Parameter container with HF XLNetModel's layout; forward(inputs_embeds) -> (hidden,)"""

    def __init__(self, config: XLNetConfig):
        super().__init__()
        self.config = config
        self.word_embedding = _WordEmbedding(config.vocab_size, config.d_model, config.initializer_range)
        self.mask_emb = nn.Parameter(torch.empty(1, 1, config.d_model).normal_(0, config.initializer_range))
        self.layer = nn.ModuleList([XLNetLayer(config) for _ in range(config.n_layer)])
        self._pos_cache = {}
        # Philox key of the dropout masks: `seed` (default rng.default_seed(): torch.initial_seed() + rank)
        self._drop_offset = 0     # advanced once per training forward

    config_class = XLNetConfig

    def pos_emb(self, L, device):
        key = (L, str(device))
        if key not in self._pos_cache:
            self._pos_cache[key] = relative_positional_encoding(L, self.config.d_model).to(device).contiguous()
        return self._pos_cache[key]

    def forward(self, inputs_embeds=None, key_len=None, **kwargs):
        """key_len: optional int32 [B] of valid key counts (opt-in padding mask, TransformerBlock(mask_padding=True));
        None reproduces the reference, which passes no attention mask."""
        cfg = self.config
        _join_weight_gradient_streams()      # leftovers of a backward pass that raised (no-op normally)
        ops.xlnet_clear_cu_budget()          # likewise: a data-parallel backward that raised before its reducer cleared it
        B, L, D = inputs_embeds.shape
        if D != cfg.d_model:
            raise ValueError(f"inputs_embeds last dim {D} != d_model {cfg.d_model}")
        pos = self.pos_emb(L, inputs_embeds.device)
        # dropout is active iff the module is in training mode, as in HF (nn.Dropout)
        p = float(cfg.dropout) if self.training else 0.0
        offset = 0
        if p > 0:
            self._drop_offset += 1
            offset = self._drop_offset
        h = inputs_embeds
        # the model-level INPUT dropout (HF :1116) rides in the first layer's kernels (csrc/xlnet_layer.hip: T4R_LAYER_FUSE_INPUT: the
        # attention-block kernel masks h on load and keeps the dropped rows for the backward, xlnet_dh masks d h) where the
        # one-kernel attention forward takes the shape; else it is the element-wise launch
        fuse_in = (p > 0 and torch.is_grad_enabled() and _FUSE_INPUT and _FUSED_ON and ops.xlnet_fused_supported(D)
                   and ops.xlnet_attn_block_supported(L, D, cfg.n_head) and key_len is None)
        if p > 0 and not fuse_in:
            h = _DropoutFn.apply(h, p, self.seed, ops.dropout_ctr_hi(offset, 255, ops.SITE_INPUT))
        # dropout(pos_emb) is drawn once per forward and shared by the layers (HF :1143).  With the stack prologue the projection
        # kernel that makes every layer's k_r also makes the dropped rows (csrc/xlnet_fused_attn.hip: GenRows): no launch of its own
        will_prepare = (torch.is_grad_enabled() or p > 0) and _STACK_PROLOGUE and ops.xlnet_fused_supported(D) and len(self.layer) > 1
        gen_pos = p > 0 and will_prepare and _GEN_POS
        if gen_pos:
            pos_b = torch.empty((B * pos.shape[0], D), device=pos.device, dtype=torch.float32)
        else:
            pos_b = ops.xlnet_pos_emb_dropout(pos, B, p, self.seed, offset) if p > 0 else None
        # no gradient wanted (evaluation / inference under torch.no_grad()): the layers run as REGISTERED operators
        # (torch.ops.t4r_hip.xlnet_layer_infer, torch_ops.py), so the body shows up in make_fx / export / compile graphs
        infer = p == 0 and not torch.is_grad_enabled()
        if infer:
            from . import torch_ops  # noqa: F401  (registers the t4r_hip library)
        # the prologue of the stack in two launches: every layer's weight planes and k_r (csrc/xlnet_fused_attn.hip)
        ws_all = None
        if not infer and _STACK_PROLOGUE and ops.xlnet_fused_supported(D) and len(self.layer) > 1:
            ws_all = ops.xlnet_stack_prepare([[q.detach() for q in layer.ordered_params()] for layer in self.layer],
                                             B, L, cfg.n_head, pos if (gen_pos or p == 0) else pos_b, p > 0,
                                             pos_dropout=(p, self.seed, offset, pos_b) if gen_pos else None)
        # the model-level OUTPUT dropout (HF :1177) rides in the last layer's feed-forward kernels (csrc/xlnet_layer.hip:
        # T4R_LAYER_FUSE_FINAL) instead of two element-wise launches over [B L, D] around the stack
        fuse_final = p > 0 and not infer and _FUSE_FINAL and _FUSED_ON and ops.xlnet_fused_supported(D)
        for i, layer in enumerate(self.layer):
            if infer:
                h = torch.ops.t4r_hip.xlnet_layer_infer(h.reshape(B * L, D), pos, layer.ordered_params(), B, L, cfg.n_head,
                                                        cfg.layer_norm_eps, key_len).view(B, L, D)
                continue
            h = _XLNetLayerFn.apply(h, layer.rel_attn.q, layer, pos, cfg.n_head, cfg.layer_norm_eps,
                                    (p, self.seed, offset, i | (ops.LAYER_FUSE_FINAL if fuse_final and i == len(self.layer) - 1 else 0)
                                     | (ops.LAYER_FUSE_INPUT if fuse_in and i == 0 else 0)),
                                    key_len, pos_b, None if ws_all is None else ws_all[i])
        if p > 0 and not fuse_final:
            h = _DropoutFn.apply(h, p, self.seed, ops.dropout_ctr_hi(offset, 255, ops.SITE_FINAL))
        return (h,)


# one wave per row block up to 64 positions (xlnet_attn.hip and the MFMA / one-kernel forms); beyond that the general kernels of
# csrc/xlnet_attn_long.hip.  What bounds the sequence now is the target kernel (csrc/masking.hip: 1023 positions).
XLNET_MAX_SEQ = 1023


class TransformerBlock(nn.Module):
    """Drop-in for tr.TransformerBlock (block/transformer.py:76-206) with the XLNet body on HIP."""

    def __init__(self, transformer, masking: Optional[MaskSequence] = None, prepare_module=None,
                 mask_padding: bool = False):
        """mask_padding (beyond the reference's signature, default off = the reference's arithmetic): the attention
        ignores padded keys as the HF body does when it is given an attention_mask (XLNet: score -1e30 except on the
        diagonal; GPT-2 / BERT: masked for every query);
        the key counts come from the item ids the masking module saw (non-pad positions, + the [MASK] slot at
        MLM inference)."""
        super().__init__()
        self.mask_padding = bool(mask_padding)
        from .transformer_hf import BertConfig, BertModel, GPT2Config, GPT2Model

        if isinstance(transformer, (XLNetConfig, GPT2Config, BertConfig)):
            self.transformer = transformer.to_huggingface_torch_model()
        elif isinstance(transformer, (XLNetModel, GPT2Model, BertModel)):
            self.transformer = transformer
        else:
            raise TypeError("TransformerBlock on the HIP path takes an XLNet / GPT-2 / BERT config or model")
        # MappingTransformerMasking (torch/utils/torch_utils.py:441-473)
        allowed = {XLNetModel: (MaskedLanguageModeling, CausalLanguageModeling),
                   GPT2Model: (CausalLanguageModeling,), BertModel: (MaskedLanguageModeling,)}[type(self.transformer)]
        if masking is not None and not isinstance(masking, allowed):
            raise ValueError(f"{masking.__class__.__name__} is not supported by: "
                             f"the {self.transformer.config_class.__name__} architecture")
        self.masking = masking
        self.prepare_module = None
        # the target kernel handles at most 1023 positions (csrc/masking.hip), and MLM
        # inference runs the body on L + 1 positions (masking.py:406-418).  Fail at construction, not at the
        # first inference call.
        tsl = getattr(self.transformer.config, "total_seq_length", None)
        if isinstance(self.transformer, XLNetModel) and tsl is not None:
            need = tsl + (1 if isinstance(masking, MaskedLanguageModeling) else 0)
            if need > XLNET_MAX_SEQ:
                raise ValueError(
                    f"XLNet on the HIP path supports sequences of at most {XLNET_MAX_SEQ} positions; "
                    f"total_seq_length={tsl}" + (" (+1 for the MLM inference slot)" if need != tsl else "")
                    + " exceeds it")

    @classmethod
    def from_registry(cls, transformer: str, d_model: int, n_head: int, n_layer: int, total_seq_length: int,
                      masking: Optional[MaskSequence] = None):
        from .transformer_hf import BertConfig, GPT2Config

        reg = {"xlnet": XLNetConfig, "gtp2": GPT2Config, "gpt2": GPT2Config, "bert": BertConfig}
        if transformer not in reg:
            raise KeyError(f"{transformer} never registered with the HIP transformer registry (supported: {sorted(reg)})")
        return cls(reg[transformer].build(d_model=d_model, n_head=n_head, n_layer=n_layer,
                                          total_seq_length=total_seq_length), masking)

    def forward(self, inputs_embeds, **kwargs):
        # the reference passes inputs_embeds only for XLNet + MLM/CLM (block/transformer.py:183-199)
        if self.mask_padding:
            ids = getattr(self.masking, "last_item_ids", None) if self.masking is not None else None
            if ids is None:
                raise ValueError("mask_padding needs the masking module to have seen the item ids of this batch")
            extra = inputs_embeds.shape[1] - ids.shape[1]       # 1 on the MLM inference grid (the [MASK] slot)
            key_len = ops.session_lengths(ids.contiguous(), self.masking.padding_idx, extra)
            return self.transformer(inputs_embeds=inputs_embeds, key_len=key_len)[0]
        return self.transformer(inputs_embeds=inputs_embeds)[0]

    def _get_name(self):
        return "TansformerBlock"

    def forward_output_size(self, input_size):
        assert len(input_size) == 3
        return torch.Size([input_size[0], input_size[1], self.transformer.config.hidden_size])
