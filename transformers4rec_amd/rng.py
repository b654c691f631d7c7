"""Device-RNG bookkeeping of the hot path (MLM draws, dropout sites, swap noise).

Every random decision on the HIP path is a pure function of (seed, stream position, element):
Philox4x32-10 keyed by `seed`, counter = (position, element index).  This module owns the two
host-side pieces:

  * the DEFAULT seed: torch's global seed (`torch.manual_seed`) combined with the data-parallel rank,
    resolved lazily at the first draw -- so `torch.manual_seed(s)` reproduces a run, another `s`
    gives other masks, and the ranks of a data-parallel job never replay each other's masks
    (the reference gets the same from torch's per-process generator);
  * checkpointing of the stream positions: `get_rng_state(model)` / `set_rng_state(model, state)`.
    They are NOT part of `state_dict` on purpose: the state_dict names are a contract with the
    reference's checkpoints, an extra key would break strict loading either way.
"""
import os

import torch

_STATE_ATTRS = ("_seed", "_rng_offset", "_drop_offset", "_post_step", "_step")
_MASK63 = 0x7FFFFFFFFFFFFFFF


def default_seed(salt: int = 0) -> int:
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        rank = dist.get_rank()
    else:
        rank = int(os.environ.get("RANK", "0"))
    return (torch.initial_seed() + 0x9E3779B97F4A7C15 * (rank + 1) + 0x632BE59BD9B4E019 * salt) & _MASK63


class SeedMixin:
    """`seed` attribute that defaults (lazily) to default_seed(salt) and can be assigned."""
    _seed = None
    _seed_salt = 0

    @property
    def seed(self):
        if self._seed is None:
            self._seed = default_seed(self._seed_salt)
        return self._seed

    @seed.setter
    def seed(self, value):
        self._seed = None if value is None else int(value) & _MASK63


_POST_KEY = "__tabular_dropout_seed__"     # reserved entry: the process-wide TabularDropout key (features.post_seed)


def get_rng_state(model):
    """{module path: {attr: value}} for every module of `model` that owns a device-RNG stream, plus the
    process-wide TabularDropout key under `_POST_KEY` (resolved here, so that a resumed run whose
    torch.initial_seed() differs still replays the same masks)."""
    from . import features

    out = {_POST_KEY: {"_seed": features.post_seed()}}
    for name, m in model.named_modules():
        st = {a: getattr(m, a) for a in _STATE_ATTRS if isinstance(getattr(m, a, None), int)}
        if isinstance(m, SeedMixin):
            st["_seed"] = m.seed          # resolve the lazy default so that the resumed run replays it
        if st:
            out[name] = st
    return out


def set_rng_state(model, state):
    mods = dict(model.named_modules())
    for name, st in state.items():
        if name == _POST_KEY:
            from . import features

            features.set_post_seed(st["_seed"])
            continue
        if name not in mods:
            raise KeyError(f"set_rng_state: no module named {name!r}")
        for a, v in st.items():
            if a not in _STATE_ATTRS:
                raise KeyError(f"set_rng_state: unknown stream attribute {a!r}")
            setattr(mods[name], a, int(v))
