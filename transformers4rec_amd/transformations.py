"""Train-time input regularisers of the input block (`pre` / `post` of the feature modules).

Mirrors transformers4rec/torch/tabular/transformations.py for the three transformations the
paper configuration uses (examples/t4rec_paper_experiments/t4r_paper_repro/transf_exp_main.py:71-91):

    StochasticSwapNoise  (:29-93)    `pre`  : swaps feature values between non-padded positions
    TabularDropout       (:145-160)  `post` : nn.Dropout on every feature embedding
    TabularLayerNorm     (:96-142)   `post` : per-feature LayerNorm (eps 1e-5), "layer-norm"

They hold configuration and parameters only; the work runs in HIP (t4r_swap_noise, t4r_dropout,
t4r_add_layernorm_*), driven by features._SeqFeaturesFn.  As in the reference they follow
`nn.Module.training` (not the `training=` keyword of forward).
"""
from typing import Dict, Optional

import torch
from torch import nn

from . import ops
from .rng import SeedMixin


class _LN(nn.Module):
    """parameter holder with torch.nn.LayerNorm's state-dict names"""

    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.eps = eps
        self.normalized_shape = (dim,)


class StochasticSwapNoise(SeedMixin, nn.Module):
    """tr.StochasticSwapNoise(schema=None, pad_token=0, replacement_prob=0.1).

    Device draws: Philox(seed, (step, module, feature)), seed default rng.default_seed();
    `set_draws` injects the reference's torch.bernoulli / torch.randperm results for one forward
    (parity tests)."""
    _seed_salt = 3

    def __init__(self, schema=None, pad_token=0, replacement_prob=0.1, seed=None):
        super().__init__()
        self.schema = schema
        self.pad_token = pad_token
        self.replacement_prob = replacement_prob
        self.seed = seed
        self._step = 0
        self._draws = None

    def set_draws(self, draws):
        """draws: {(module_name, feature): (bern uint8 like the feature, perm int64 [#non-pad])}"""
        self._draws = draws

    def augment_module(self, inputs, names, item_ids, module_name, module_index):
        """-> {name: swapped tensor} for the features `names` of one feature module."""
        if not self.training:
            return {}
        out = {}
        for fi, n in enumerate(names):
            x = inputs[n]
            bern = perm = None
            if self._draws is not None:
                bern, perm = self._draws[(module_name, n)]
            ctr = ops.dropout_ctr_hi(self._step, 0xFD, module_index * 16 + fi)
            out[n] = ops.swap_noise(x.contiguous(), item_ids, self.replacement_prob, self.pad_token, bern, perm,
                                    self.seed, ctr)
        return out

    def next_step(self):
        self._step += 1

    def extra_repr(self):
        return f"replacement_prob={self.replacement_prob}, pad_token={self.pad_token}"


class TabularDropout(nn.Module):
    """tr.TabularDropout(dropout_rate)."""

    def __init__(self, dropout_rate=0.0):
        super().__init__()
        if not 0.0 <= dropout_rate < 1.0:
            raise ValueError("dropout_rate must be in [0, 1)")
        self.dropout_rate = dropout_rate

    def extra_repr(self):
        return f"p={self.dropout_rate}"


class TabularLayerNorm(nn.Module):
    """tr.TabularLayerNorm(features_dim): LayerNorm per feature; features of dim 1 are skipped
    (transformations.py:110-117)."""

    def __init__(self, features_dim: Optional[Dict[str, int]] = None):
        super().__init__()
        self.feature_layer_norm = nn.ModuleDict()
        self.build(features_dim)

    def build(self, features_dim):
        for n, d in (features_dim or {}).items():
            if d > 1 and n not in self.feature_layer_norm:
                self.feature_layer_norm[n] = _LN(d)
        return self


def parse_pre(pre):
    """-> StochasticSwapNoise or None ("stochastic-swap-noise" / "ssn" strings build a default one)."""
    if pre is None:
        return None
    items = list(pre) if isinstance(pre, (list, tuple)) else [pre]
    out = None
    for it in items:
        if isinstance(it, str):
            if it not in ("stochastic-swap-noise", "ssn"):
                raise ValueError(f"unsupported pre transformation {it!r}")
            it = StochasticSwapNoise()
        if not isinstance(it, StochasticSwapNoise):
            raise NotImplementedError(f"pre transformation {type(it).__name__} is off the hot path")
        if out is not None:
            raise NotImplementedError("one StochasticSwapNoise per module")
        out = it
    return out


def parse_post(post, features_dim):
    """-> nn.ModuleList of TabularDropout / TabularLayerNorm in application order
    ("layer-norm" / "dropout" strings are built like the reference registry does)."""
    if post is None:
        return None
    items = list(post) if isinstance(post, (list, tuple)) else [post]
    out = []
    for it in items:
        if isinstance(it, str):
            if it == "layer-norm":
                it = TabularLayerNorm()
            elif it == "dropout":
                it = TabularDropout()
            else:
                raise ValueError(f"unsupported post transformation {it!r}")
        if isinstance(it, TabularLayerNorm):
            it.build(features_dim)      # TabularModule.build -> post.build(output sizes), transformations.py:134-139
        elif not isinstance(it, TabularDropout):
            raise NotImplementedError(f"post transformation {type(it).__name__} is off the hot path")
        out.append(it)
    return nn.ModuleList(out) if out else None
