"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the device random streams of the HIP path.

The reference draws its training randomness from torch's generator (`torch.bernoulli` / `torch.multinomial` in
transformers4rec/torch/masking.py:425-459, `nn.Dropout` at the seven sites of HF modeling_xlnet.py:132,147,301,303,
1116,1143,1177).  A GPU implementation cannot replay torch's CPU generator; what it CAN do is make every random
decision a pure, documented function of (seed, stream position, element) -- and then the oracle can be driven by
exactly the decisions the device took.  This file restates those functions with integer arithmetic only (numpy
uint64), so that a whole training step -- or a 200-step trajectory -- of the CPU oracle runs in LOCKSTEP with the HIP
path at dropout 0.3 (VERDICT r5 next #1b).

What is restated (the definitions live in transformers4rec_amd/csrc/t4r_common.h and csrc/masking.hip; this file
was written from their documented contract, tests/test_round6_gpu.py checks it bit for bit against masks exported
from the device, tests/test_device_rng_oracle.py against the published Philox known-answer vectors):

  * Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123
    kat_vectors) keyed by the 64-bit seed, 128-bit counter = (ctr_lo, ctr_hi);
  * dropout site masks: element idx keeps its value iff word (idx & 3) of block (ctr_lo = idx >> 2, ctr_hi) is
    >= ceil(p * 2^24) << 8, ctr_hi = (forward offset << 16) | (layer << 8) | site;
  * MLM training draws: bernoulli word x of block (offset + b*L + l, 0); j1 / j2 selectors words x / y of block
    (offset + b, 1), index = min(n - 1, int(float32(word >> 8) * 2^-24 * n)).

`oracle/device_rng.c` is the same arithmetic in C (OpenMP) for full-size tensors; `use_c()` says whether the compiled
form is present; results are identical by test.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_MASK32 = np.uint64(0xFFFFFFFF)
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85

# dropout sites (transformers4rec_amd/ops.py: SITE_*; csrc/xlnet_layer.hip ctr_hi)
SITE_INPUT, SITE_POS, SITE_PROB, SITE_ATTN_OUT, SITE_FF_ACT, SITE_FF_OUT, SITE_FINAL = range(7)
MODEL_LEVEL = 255      # the `layer` byte of the model-level sites (input, shared pos_emb, final)


def philox4x32_10(seed, ctr_lo, ctr_hi):
    """four uint32 words [.., 4] of Philox4x32-10(key = seed, counter = (ctr_lo, ctr_hi)); ctr_lo: uint64 array."""
    ctr_lo = np.asarray(ctr_lo, dtype=np.uint64)
    ctr_hi = np.uint64(int(ctr_hi) & 0xFFFFFFFFFFFFFFFF)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    c0, c1 = ctr_lo & _MASK32, ctr_lo >> np.uint64(32)
    c2 = np.full_like(c0, ctr_hi & _MASK32)
    c3 = np.full_like(c0, ctr_hi >> np.uint64(32))
    a, b = seed & 0xFFFFFFFF, seed >> 32
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2                  # 32 x 32 -> 64 bit products: exact in uint64
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(a)
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(b)
        c0, c1, c2, c3 = n0, p1 & _MASK32, n2, p0 & _MASK32
        a, b = (a + _W0) & 0xFFFFFFFF, (b + _W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], -1).astype(np.uint32)


def dropout_ctr_hi(offset, layer, site):
    return ((int(offset) << 16) | ((int(layer) & 0xFF) << 8) | int(site)) & 0xFFFFFFFFFFFFFFFF


def keep_threshold(p):
    """keep <=> word >= thr.  thr is the integer form of `float32(word >> 8) * 2^-24 >= float32(p)`."""
    p = np.float32(p)
    if p <= 0:
        return 0
    if p >= 1:
        return 0xFFFFFFFF
    return int(np.ceil(p * np.float32(16777216.0))) << 8


def _dropout_keep_np(seed, ctr_hi, n, p):
    nblk = (n + 3) // 4
    w = philox4x32_10(seed, np.arange(nblk, dtype=np.uint64), ctr_hi).reshape(-1)[:n]
    return (w >= np.uint32(keep_threshold(p))).astype(np.uint8)


def _mlm_draws_np(seed, offset, B, L, p):
    w = philox4x32_10(seed, np.uint64(offset) + np.arange(B * L, dtype=np.uint64), 0)[:, 0]
    unit = (w >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    bern = (unit < np.float32(p)).reshape(B, L).astype(np.uint8)
    r = philox4x32_10(seed, np.uint64(offset) + np.arange(B, dtype=np.uint64), 1)
    u = (r[:, :2] >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return bern, u[:, 0].copy(), u[:, 1].copy()


# ---------------------------------------------------------------------------------------------- C form
_C = None


def _load_c():
    global _C
    if _C is None:
        path = os.path.join(HERE, "_build", "libt4r_oracle_rng.so")
        if not os.path.exists(path):
            _C = False
        else:
            lib = ctypes.CDLL(path)
            lib.t4r_oracle_dropout_keep.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                                                    ctypes.c_void_p]
            lib.t4r_oracle_mlm_draws.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64,
                                                 ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            lib.t4r_oracle_philox.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
            _C = lib
    return _C


def use_c():
    return bool(_load_c())


def dropout_keep(seed, ctr_hi, n, p, force_numpy=False):
    """uint8 [n]: 1 where element idx of the dropout site (seed, ctr_hi) keeps its value"""
    lib = None if force_numpy else _load_c()
    if not lib:
        return _dropout_keep_np(seed, ctr_hi, int(n), p)
    out = np.empty(int(n), dtype=np.uint8)
    lib.t4r_oracle_dropout_keep(int(seed) & 0xFFFFFFFFFFFFFFFF, int(ctr_hi) & 0xFFFFFFFFFFFFFFFF, int(n), keep_threshold(p),
                                out.ctypes.data)
    return out


def mlm_draws(seed, offset, B, L, p, force_numpy=False):
    """(bern uint8 [B, L], u1 float32 [B], u2 float32 [B]) of one MLM training call at stream position `offset`"""
    lib = None if force_numpy else _load_c()
    if not lib:
        return _mlm_draws_np(seed, int(offset), int(B), int(L), p)
    bern = np.empty((B, L), dtype=np.uint8)
    u1, u2 = np.empty(B, dtype=np.float32), np.empty(B, dtype=np.float32)
    lib.t4r_oracle_mlm_draws(int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset), int(B), int(L), float(p), bern.ctypes.data,
                             u1.ctypes.data, u2.ctypes.data)
    return bern, u1, u2


# ---------------------------------------------------------------------------------------------- users
def mlm_targets_train_device(ids, seed, offset, p, padding_idx=0):
    """(mask_schema bool [B, L], labels int64 [B, L]) the device draws for `ids` (torch int64 [B, L]) at stream position
    `offset`: the reference's rule (masking.py:425-459, restated in t4r_oracle.mlm_targets_train) driven by the device's
    draws -- j1 is the floor(u1 * n_nonpad)-th non-pad position, j2 the floor(u2 * n_labels)-th labelled one."""
    import torch

    B, L = ids.shape
    bern, u1, u2 = mlm_draws(seed, offset, B, L, p)
    idn = ids.numpy()
    nonpad = idn != padding_idx
    lab = (bern != 0) & nonpad
    n_np = nonpad.sum(1)
    rows = np.nonzero(n_np > 0)[0]

    def kth(bits, k):                      # position of the k-th (0-based) set bit of every row
        c = np.cumsum(bits, 1)
        return np.argmax((c == (k + 1)[:, None]) & bits, 1)

    k1 = np.minimum(n_np[rows] - 1, (u1[rows] * n_np[rows].astype(np.float32)).astype(np.int64))
    j1 = kth(nonpad[rows], k1)
    lab[rows, j1] = True
    n_lab = lab.sum(1)
    full = rows[n_lab[rows] == n_np[rows]]
    if full.size:
        k2 = np.minimum(n_lab[full] - 1, (u2[full] * n_lab[full].astype(np.float32)).astype(np.int64))
        j2 = kth(lab[full], k2)
        lab[full, j2] = False
    labels = np.where(lab, idn, padding_idx)
    return torch.from_numpy(lab.copy()), torch.from_numpy(labels.astype(np.int64))


def xlnet_dropout_masks(B, L, D, n_head, n_layer, p, seed, offset, d_inner=None):
    """every dropout mask of ONE training forward of the XLNet body as the device draws them (`offset` = the model's
    forward counter, 1 for the first training forward): dict input [B,L,D], pos [B,2L,D] (drawn once, shared by the
    layers: HF :1143), layers = list of dict prob [B,n,L,L] / attn_out [B,L,D] / ff_act [B,L,4D] / ff_out [B,L,D],
    final [B,L,D]; uint8 torch tensors (1 = kept).  Element index = the row-major index in the shapes above."""
    import torch

    F = 4 * D if d_inner is None else d_inner

    def site(shape, layer, s):
        n = int(np.prod(shape))
        return torch.from_numpy(dropout_keep(seed, dropout_ctr_hi(offset, layer, s), n, p)).view(*shape)

    return dict(input=site((B, L, D), MODEL_LEVEL, SITE_INPUT), pos=site((B, 2 * L, D), MODEL_LEVEL, SITE_POS),
                layers=[dict(prob=site((B, n_head, L, L), i, SITE_PROB), attn_out=site((B, L, D), i, SITE_ATTN_OUT),
                             ff_act=site((B, L, F), i, SITE_FF_ACT), ff_out=site((B, L, D), i, SITE_FF_OUT))
                        for i in range(n_layer)],
                final=site((B, L, D), MODEL_LEVEL, SITE_FINAL))
