"""GPT-2 and BERT transformer bodies for TransformerBlock.

Host-side mirror of what the reference instantiates through
  GPT2Config.build   transformers4rec/config/transformer.py:218-260   (+ GPT2Prepare, block/transformer.py:55-73)
  BertConfig.build   transformers4rec/config/transformer.py:493-534
i.e. HF `GPT2Model` / `BertModel` (third-party; gpt2/modeling_gpt2.py, bert/modeling_bert.py) fed with
`inputs_embeds` only.  Parameter containers use HF's state_dict names so reference checkpoints load.
Quirks of the reference builders are frozen here on purpose:
  * GPT-2: `layer_norm_eps` is ignored by HF (it reads `layer_norm_epsilon`) -> effective eps 1e-5;
    the tril `head_mask` of GPT2Prepare is dropped by HF 5.x (and was a no-op before);
  * BERT: `dropout` is ignored (HF defaults 0.1 / 0.1 stay), `intermediate_size` stays 3072,
    `max_position_embeddings = total_seq_length + 2`; the pooler is computed by HF and discarded by
    TransformerBlock, so it is never evaluated here (its parameters exist for checkpoint parity).
Arithmetic: csrc/gemm_f32.hip (Conv1D / Linear with fused bias, GELU, dropout + residual epilogues),
csrc/mha.hip (attention core), csrc/elementwise.hip (LayerNorm, dropout, position embeddings).
"""
import math
from dataclasses import dataclass
from typing import Optional

import torch
from torch import nn

from . import ops
from .masking import _grad_buf
from .rng import SeedMixin

S_IN, S_PROB, S_AO, S_FF, S_FO, S_FINAL = (ops.SITE_INPUT, ops.SITE_PROB, ops.SITE_ATTN_OUT, ops.SITE_FF_ACT,
                                           ops.SITE_FF_OUT, ops.SITE_FINAL)


def _drop(p, seed, offset, layer, site):
    return (p, seed, ops.dropout_ctr_hi(offset, layer, site)) if p > 0 else ops.NO_DROP


# ====================================================================================== configs
@dataclass
class GPT2Config:
    n_embd: int = 256
    n_head: int = 4
    n_layer: int = 6
    n_inner: int = 1024
    n_positions: int = 50
    activation_function: str = "gelu"
    initializer_range: float = 0.01
    layer_norm_epsilon: float = 1e-5
    resid_pdrop: float = 0.3
    embd_pdrop: float = 0.3
    attn_pdrop: float = 0.3
    vocab_size: int = 1
    model_type: str = "gpt2"

    @classmethod
    def build(cls, d_model, n_head, n_layer, total_seq_length, hidden_act="gelu", initializer_range=0.01,
              layer_norm_eps=0.03, dropout=0.3, pad_token=0, log_attention_weights=False, **kwargs):
        if hidden_act != "gelu":
            raise NotImplementedError("only hidden_act='gelu' (erf) is on the hot path")
        # layer_norm_eps is accepted and -- exactly like the reference -- has no effect on GPT-2
        return cls(n_embd=d_model, n_inner=4 * d_model, n_layer=n_layer, n_head=n_head,
                   activation_function=hidden_act, initializer_range=initializer_range, resid_pdrop=dropout,
                   embd_pdrop=dropout, attn_pdrop=dropout, n_positions=total_seq_length)

    @property
    def hidden_size(self):
        return self.n_embd

    def to_huggingface_torch_model(self):
        return GPT2Model(self)

    def to_torch_model(self, input_features, *prediction_task, **kwargs):
        from .model import Model
        from .transformer import TransformerBlock

        return Model(input_features, TransformerBlock(self, masking=input_features.masking), prediction_task[0])


@dataclass
class BertConfig:
    hidden_size: int = 512
    num_hidden_layers: int = 12
    num_attention_heads: int = 8
    intermediate_size: int = 3072
    max_position_embeddings: int = 102
    type_vocab_size: int = 2
    hidden_act: str = "gelu"
    initializer_range: float = 0.01
    layer_norm_eps: float = 0.03
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    vocab_size: int = 1
    model_type: str = "bert"

    @classmethod
    def build(cls, d_model, n_head, n_layer, total_seq_length, hidden_act="gelu", initializer_range=0.01,
              layer_norm_eps=0.03, dropout=0.3, pad_token=0, log_attention_weights=False, **kwargs):
        if hidden_act != "gelu":
            raise NotImplementedError("only hidden_act='gelu' (erf) is on the hot path")
        # `dropout` is passed to HF as an unknown kwarg by the reference and ignored: 0.1 / 0.1 stay
        return cls(hidden_size=d_model, num_hidden_layers=n_layer, num_attention_heads=n_head,
                   hidden_act=hidden_act, initializer_range=initializer_range, layer_norm_eps=layer_norm_eps,
                   max_position_embeddings=total_seq_length + 2,
                   hidden_dropout_prob=kwargs.get("hidden_dropout_prob", 0.1),
                   attention_probs_dropout_prob=kwargs.get("attention_probs_dropout_prob", 0.1),
                   intermediate_size=kwargs.get("intermediate_size", 3072))

    def to_huggingface_torch_model(self):
        return BertModel(self)

    def to_torch_model(self, input_features, *prediction_task, **kwargs):
        from .model import Model
        from .transformer import TransformerBlock

        return Model(input_features, TransformerBlock(self, masking=input_features.masking), prediction_task[0])


# ====================================================================================== parameter holders
class _LN(nn.Module):
    def __init__(self, dim, eps):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _Conv1D(nn.Module):
    """HF Conv1D: weight [in, out], y = x @ W + b"""

    def __init__(self, nin, nout, std):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(nin, nout).normal_(0, std))
        self.bias = nn.Parameter(torch.zeros(nout))


class _Linear(nn.Module):
    """torch Linear: weight [out, in]"""

    def __init__(self, nin, nout, std):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(nout, nin).normal_(0, std))
        self.bias = nn.Parameter(torch.zeros(nout))


class _Emb(nn.Module):
    def __init__(self, n, dim, std):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, dim).normal_(0, std))


# ====================================================================================== GPT-2
class _GPT2Attn(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        D = cfg.n_embd
        self.c_attn = _Conv1D(D, 3 * D, cfg.initializer_range)
        self.c_proj = _Conv1D(D, D, cfg.initializer_range / math.sqrt(2 * cfg.n_layer))


class _GPT2MLP(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        D = cfg.n_embd
        self.c_fc = _Conv1D(D, cfg.n_inner, cfg.initializer_range)
        self.c_proj = _Conv1D(cfg.n_inner, D, cfg.initializer_range / math.sqrt(2 * cfg.n_layer))


class GPT2Block(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.ln_1 = _LN(cfg.n_embd, cfg.layer_norm_epsilon)
        self.attn = _GPT2Attn(cfg)
        self.ln_2 = _LN(cfg.n_embd, cfg.layer_norm_epsilon)
        self.mlp = _GPT2MLP(cfg)


class _GPT2BlockFn(torch.autograd.Function):
    """HF GPT2Block.forward (gpt2 :262-309): pre-LN causal attention + pre-LN GELU MLP, residual adds."""

    @staticmethod
    def forward(ctx, h, anchor, blk, n_head, drop, key_len=None):
        B, L, D = h.shape
        T = B * L
        p, seed, off, li = drop
        ctx.key_len = key_len
        h2 = h.contiguous().view(T, D)
        a, m = blk.attn, blk.mlp
        x1, mean1, rstd1 = ops.add_layernorm_fwd(h2, None, blk.ln_1.weight.detach(), blk.ln_1.bias.detach(), blk.ln_1.eps)
        qkv = ops.gemm(x1, a.c_attn.weight.detach(), bias=a.c_attn.bias.detach(), epilogue=ops.EPI_BIAS)
        att, lse = ops.mha_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, L, n_head, True,
                               _drop(p, seed, off, li, S_PROB), key_len=key_len)
        h1 = ops.gemm(att, a.c_proj.weight.detach(), bias=a.c_proj.bias.detach(), epilogue=ops.EPI_BIAS_RESID,
                      aux=h2, drop=_drop(p, seed, off, li, S_AO))
        x2, mean2, rstd2 = ops.add_layernorm_fwd(h1, None, blk.ln_2.weight.detach(), blk.ln_2.bias.detach(), blk.ln_2.eps)
        fpre = torch.empty((T, m.c_fc.weight.shape[1]), device=h.device, dtype=torch.float32)
        fact = ops.gemm(x2, m.c_fc.weight.detach(), bias=m.c_fc.bias.detach(), epilogue=ops.EPI_BIAS_GELU, aux=fpre)
        out = ops.gemm(fact, m.c_proj.weight.detach(), bias=m.c_proj.bias.detach(), epilogue=ops.EPI_BIAS_RESID,
                       aux=h1, drop=_drop(p, seed, off, li, S_FO))
        ctx.blk, ctx.cfg = blk, (B, L, D, n_head, drop)
        ctx.save_for_backward(h2, mean1, rstd1, x1, qkv, att, lse, h1, mean2, rstd2, x2, fpre, fact)
        return out.view(B, L, D)

    @staticmethod
    def backward(ctx, dout):
        h2, mean1, rstd1, x1, qkv, att, lse, h1, mean2, rstd2, x2, fpre, fact = ctx.saved_tensors
        blk = ctx.blk
        B, L, D, n_head, (p, seed, off, li) = ctx.cfg
        T = B * L
        a, m = blk.attn, blk.mlp
        dh1 = dout.contiguous().view(T, D).clone()        # residual stream gradient (accumulated below)
        dm = ops.dropout(dh1.view(-1), p, seed, ops.dropout_ctr_hi(off, li, S_FO)).view(T, D) if p > 0 else dh1
        dfact = ops.gemm(dm, m.c_proj.weight.detach(), False, True)
        ops.gemm(fact, dm, True, False, splitk=-1, accumulate=True, out=_grad_buf(m.c_proj.weight))
        ops.colsum_(dm, _grad_buf(m.c_proj.bias))
        ops.act_bwd_bias(dfact, fpre, _grad_buf(m.c_fc.bias), 0)
        dx2 = ops.gemm(dfact, m.c_fc.weight.detach(), False, True)
        ops.gemm(x2, dfact, True, False, splitk=-1, accumulate=True, out=_grad_buf(m.c_fc.weight))
        ops.add_layernorm_bwd(h1, None, blk.ln_2.weight.detach(), mean2, rstd2, dx2, _grad_buf(blk.ln_2.weight),
                              _grad_buf(blk.ln_2.bias), dx=dh1, accumulate_dx=True)
        dh = dh1                                           # d h = d h1 (residual) + LN1 path
        da = ops.dropout(dh1.view(-1), p, seed, ops.dropout_ctr_hi(off, li, S_AO)).view(T, D) if p > 0 else dh1
        datt = ops.gemm(da, a.c_proj.weight.detach(), False, True)
        ops.gemm(att, da, True, False, splitk=-1, accumulate=True, out=_grad_buf(a.c_proj.weight))
        ops.colsum_(da, _grad_buf(a.c_proj.bias))
        dqkv = ops.mha_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], att, lse, datt, B, L, n_head, True,
                           _drop(p, seed, off, li, S_PROB), fused_out=True, key_len=ctx.key_len)
        dx1 = ops.gemm(dqkv, a.c_attn.weight.detach(), False, True)
        ops.gemm(x1, dqkv, True, False, splitk=-1, accumulate=True, out=_grad_buf(a.c_attn.weight))
        ops.colsum_(dqkv, _grad_buf(a.c_attn.bias))
        ops.add_layernorm_bwd(h2, None, blk.ln_1.weight.detach(), mean1, rstd1, dx1, _grad_buf(blk.ln_1.weight),
                              _grad_buf(blk.ln_1.bias), dx=dh, accumulate_dx=True)
        return dh.view(B, L, D), None, None, None, None, None


class _PosEmbFn(torch.autograd.Function):
    """x + pos[0:L] (+ token_type[0]) -- learned absolute positions (GPT-2 wpe / BERT embeddings)"""

    @staticmethod
    def forward(ctx, x, pos, tt):
        ctx.pos, ctx.tt, ctx.L = pos, tt, x.shape[1]
        return ops.add_pos_fwd(x.contiguous(), pos.detach()[: x.shape[1]].contiguous(),
                               None if tt is None else tt.detach()[0].contiguous())

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        ops.add_pos_bwd_(dy, _grad_buf(ctx.pos)[: ctx.L])
        if ctx.tt is not None:
            ops.colsum_(dy.view(-1, dy.shape[-1]), _grad_buf(ctx.tt)[0])
        return dy, None, None


class _LNFn(torch.autograd.Function):
    """plain LayerNorm over the last dim (GPT-2 ln_f, BERT embeddings.LayerNorm)"""

    @staticmethod
    def forward(ctx, x, ln):
        shp = x.shape
        x2 = x.contiguous().view(-1, shp[-1])
        y, mean, rstd = ops.add_layernorm_fwd(x2, None, ln.weight.detach(), ln.bias.detach(), ln.eps)
        ctx.ln = ln
        ctx.save_for_backward(x2, mean, rstd)
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd = ctx.saved_tensors
        ln = ctx.ln
        dx = ops.add_layernorm_bwd(x2, None, ln.weight.detach(), mean, rstd, dy.contiguous().view(x2.shape),
                                   _grad_buf(ln.weight), _grad_buf(ln.bias))
        return dx.view(dy.shape), None


class _DropFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, ctr):
        ctx.cfg = (p, seed, ctr)
        return ops.dropout(x.contiguous().view(-1), p, seed, ctr).view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        p, seed, ctr = ctx.cfg
        return ops.dropout(dy.contiguous().view(-1), p, seed, ctr).view(dy.shape), None, None, None


class GPT2Model(SeedMixin, nn.Module):
    config_class = GPT2Config

    def __init__(self, config: GPT2Config):
        super().__init__()
        if config.resid_pdrop != config.attn_pdrop or config.resid_pdrop != config.embd_pdrop:
            raise NotImplementedError("one dropout rate for embd/attn/resid (what GPT2Config.build sets)")
        self.config = config
        self.wte = _Emb(config.vocab_size, config.n_embd, config.initializer_range)   # unused with inputs_embeds
        self.wpe = _Emb(config.n_positions, config.n_embd, config.initializer_range)
        self.h = nn.ModuleList([GPT2Block(config) for _ in range(config.n_layer)])
        self.ln_f = _LN(config.n_embd, config.layer_norm_epsilon)
        self._drop_offset = 0      # `seed`: rng.SeedMixin (torch.initial_seed() + rank unless assigned)

    def forward(self, inputs_embeds=None, key_len=None, **kwargs):
        cfg = self.config
        B, L, D = inputs_embeds.shape
        if L > cfg.n_positions:
            raise ValueError(f"sequence length {L} > n_positions {cfg.n_positions}")
        p = float(cfg.resid_pdrop) if self.training else 0.0
        off = 0
        if p > 0:
            self._drop_offset += 1
            off = self._drop_offset
        h = _PosEmbFn.apply(inputs_embeds, self.wpe.weight, None)
        if p > 0:
            h = _DropFn.apply(h, p, self.seed, ops.dropout_ctr_hi(off, 255, S_IN))
        for i, blk in enumerate(self.h):
            h = _GPT2BlockFn.apply(h, blk.ln_1.weight, blk, cfg.n_head, (p, self.seed, off, i), key_len)
        return (_LNFn.apply(h, self.ln_f),)


# ====================================================================================== BERT
class _BertSelf(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        D, s = cfg.hidden_size, cfg.initializer_range
        self.query, self.key, self.value = _Linear(D, D, s), _Linear(D, D, s), _Linear(D, D, s)


class _BertSelfOutput(nn.Module):
    def __init__(self, cfg, nin):
        super().__init__()
        self.dense = _Linear(nin, cfg.hidden_size, cfg.initializer_range)
        self.LayerNorm = _LN(cfg.hidden_size, cfg.layer_norm_eps)


class _BertAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.self = _BertSelf(cfg)
        self.output = _BertSelfOutput(cfg, cfg.hidden_size)


class _BertIntermediate(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dense = _Linear(cfg.hidden_size, cfg.intermediate_size, cfg.initializer_range)


class BertLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.attention = _BertAttention(cfg)
        self.intermediate = _BertIntermediate(cfg)
        self.output = _BertSelfOutput(cfg, cfg.intermediate_size)


class _BertLayerFn(torch.autograd.Function):
    """HF BertLayer.forward: post-LN bidirectional attention + GELU MLP (no attention mask)."""

    @staticmethod
    def forward(ctx, h, anchor, lay, n_head, drop, key_len=None):
        B, L, D = h.shape
        T = B * L
        ph, pa, seed, off, li = drop
        ctx.key_len = key_len
        h2 = h.contiguous().view(T, D)
        s, so, it, o = lay.attention.self, lay.attention.output, lay.intermediate, lay.output
        lin = lambda x, l: ops.gemm(x, l.weight.detach(), False, True, bias=l.bias.detach(), epilogue=ops.EPI_BIAS)
        q, k, v = lin(h2, s.query), lin(h2, s.key), lin(h2, s.value)
        att, lse = ops.mha_fwd(q, k, v, B, L, n_head, False, _drop(pa, seed, off, li, S_PROB), key_len=key_len)
        ao = lin(att, so.dense)
        h1, mean1, rstd1 = ops.add_layernorm_fwd(ao, h2, so.LayerNorm.weight.detach(), so.LayerNorm.bias.detach(),
                                                 so.LayerNorm.eps, _drop(ph, seed, off, li, S_AO))
        ipre = torch.empty((T, it.dense.weight.shape[0]), device=h.device, dtype=torch.float32)
        iact = ops.gemm(h1, it.dense.weight.detach(), False, True, bias=it.dense.bias.detach(),
                        epilogue=ops.EPI_BIAS_GELU, aux=ipre)
        oo = lin(iact, o.dense)
        out, mean2, rstd2 = ops.add_layernorm_fwd(oo, h1, o.LayerNorm.weight.detach(), o.LayerNorm.bias.detach(),
                                                  o.LayerNorm.eps, _drop(ph, seed, off, li, S_FO))
        ctx.lay, ctx.cfg = lay, (B, L, D, n_head, drop)
        ctx.save_for_backward(h2, q, k, v, att, lse, ao, mean1, rstd1, h1, ipre, iact, oo, mean2, rstd2)
        return out.view(B, L, D)

    @staticmethod
    def backward(ctx, dout):
        h2, q, k, v, att, lse, ao, mean1, rstd1, h1, ipre, iact, oo, mean2, rstd2 = ctx.saved_tensors
        lay = ctx.lay
        B, L, D, n_head, (ph, pa, seed, off, li) = ctx.cfg
        T = B * L
        s, so, it, o = lay.attention.self, lay.attention.output, lay.intermediate, lay.output
        dy = dout.contiguous().view(T, D)
        r = ops.add_layernorm_bwd(oo, h1, o.LayerNorm.weight.detach(), mean2, rstd2, dy, _grad_buf(o.LayerNorm.weight),
                                  _grad_buf(o.LayerNorm.bias), drop=_drop(ph, seed, off, li, S_FO))
        dh1, doo = r if isinstance(r, tuple) else (r, r)
        diact = ops.gemm(doo, o.dense.weight.detach(), False, False)
        ops.gemm(doo, iact, True, False, splitk=-1, accumulate=True, out=_grad_buf(o.dense.weight))
        ops.colsum_(doo, _grad_buf(o.dense.bias))
        ops.act_bwd_bias(diact, ipre, _grad_buf(it.dense.bias), 0)
        ops.gemm(diact, it.dense.weight.detach(), False, False, accumulate=True, out=dh1)      # d h1 += ...
        ops.gemm(diact, h1, True, False, splitk=-1, accumulate=True, out=_grad_buf(it.dense.weight))
        r = ops.add_layernorm_bwd(ao, h2, so.LayerNorm.weight.detach(), mean1, rstd1, dh1, _grad_buf(so.LayerNorm.weight),
                                  _grad_buf(so.LayerNorm.bias), drop=_drop(ph, seed, off, li, S_AO))
        dh, dao = r if isinstance(r, tuple) else (r, r)
        datt = ops.gemm(dao, so.dense.weight.detach(), False, False)
        ops.gemm(dao, att, True, False, splitk=-1, accumulate=True, out=_grad_buf(so.dense.weight))
        ops.colsum_(dao, _grad_buf(so.dense.bias))
        dq, dk, dv = ops.mha_bwd(q, k, v, att, lse, datt, B, L, n_head, False, _drop(pa, seed, off, li, S_PROB),
                                 key_len=ctx.key_len)
        if dh is dao:
            dh = dh.clone()
        for g, l in ((dq, s.query), (dk, s.key), (dv, s.value)):
            ops.gemm(g, l.weight.detach(), False, False, accumulate=True, out=dh)
            ops.gemm(g, h2, True, False, splitk=-1, accumulate=True, out=_grad_buf(l.weight))
            ops.colsum_(g, _grad_buf(l.bias))
        return dh.view(B, L, D), None, None, None, None, None


class _BertEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        s = cfg.initializer_range
        self.word_embeddings = _Emb(cfg.vocab_size, cfg.hidden_size, s)               # unused with inputs_embeds
        self.position_embeddings = _Emb(cfg.max_position_embeddings, cfg.hidden_size, s)
        self.token_type_embeddings = _Emb(cfg.type_vocab_size, cfg.hidden_size, s)
        self.LayerNorm = _LN(cfg.hidden_size, cfg.layer_norm_eps)


class _BertEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(cfg) for _ in range(cfg.num_hidden_layers)])


class _BertPooler(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dense = _Linear(cfg.hidden_size, cfg.hidden_size, cfg.initializer_range)


class BertModel(SeedMixin, nn.Module):
    config_class = BertConfig

    def __init__(self, config: BertConfig):
        super().__init__()
        self.config = config
        self.embeddings = _BertEmbeddings(config)
        self.encoder = _BertEncoder(config)
        self.pooler = _BertPooler(config)      # HF computes it, TransformerBlock drops it: never evaluated
        self._drop_offset = 0      # `seed`: rng.SeedMixin (torch.initial_seed() + rank unless assigned)

    def forward(self, inputs_embeds=None, key_len=None, **kwargs):
        cfg = self.config
        B, L, D = inputs_embeds.shape
        if L > cfg.max_position_embeddings:
            raise ValueError(f"sequence length {L} > max_position_embeddings {cfg.max_position_embeddings}")
        ph = float(cfg.hidden_dropout_prob) if self.training else 0.0
        pa = float(cfg.attention_probs_dropout_prob) if self.training else 0.0
        off = 0
        if ph > 0 or pa > 0:
            self._drop_offset += 1
            off = self._drop_offset
        e = self.embeddings
        h = _PosEmbFn.apply(inputs_embeds, e.position_embeddings.weight, e.token_type_embeddings.weight)
        h = _LNFn.apply(h, e.LayerNorm)
        if ph > 0:
            h = _DropFn.apply(h, ph, self.seed, ops.dropout_ctr_hi(off, 255, S_IN))
        for i, lay in enumerate(self.encoder.layer):
            h = _BertLayerFn.apply(h, lay.attention.self.query.weight, lay, cfg.num_attention_heads,
                                   (ph, pa, self.seed, off, i), key_len)
        return (h,)
