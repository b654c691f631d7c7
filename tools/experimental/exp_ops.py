"""Python wrappers of the entry points that exist only in the A/B variant library (build_variant.sh, -DT4R_EXPERIMENTAL),
loaded through T4R_HIP_LIB like any other build of the library."""
import ctypes
import os

import torch

from transformers4rec_amd import _lib
from transformers4rec_amd.ops import _chk, _p, _stream


def available():
    path = os.environ.get("T4R_HIP_LIB")
    if not path or not os.path.exists(path):
        return False
    return hasattr(ctypes.CDLL(path), "t4r_xlnet_attn_block_bwd")


def _raw():
    lib = ctypes.CDLL(os.environ["T4R_HIP_LIB"])
    lib.t4r_xlnet_attn_block_bwd_part_floats.restype = ctypes.c_long
    lib.t4r_xlnet_attn_block_bwd_part_floats.argtypes = [ctypes.c_int] * 4
    f = lib.t4r_xlnet_attn_block_bwd
    f.restype = ctypes.c_int
    P, I, L_, F, Q = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_ulonglong
    f.argtypes = [P] + [P] * 6 + [P] * 4 + [P, P] + [L_] + [P] * 3 + [P] * 4 + [P] * 5 + [I] * 4 + [F, Q, Q, Q] + [P]
    return lib


def xlnet_attn_block_bwd(dy, saved, h, planes, wq, wk, wv, kr, r_w_bias, r_r_bias, gamma, d_rw, d_rr, d_gamma, d_beta, B, L, n_head,
                         drop_p=0.0, seed=0, ctr_prob=0, ctr_out=0, key_len=None):
    """backward of xlnet_attn_block_fwd: -> dh, dao, dqkv [3, T, D], dkr (shape of kr); d_rw / d_rr / d_gamma / d_beta accumulated"""
    lib = _raw()
    T, D = dy.shape
    dev = dy.device
    dh, dao = torch.empty((T, D), device=dev), torch.empty((T, D), device=dev)
    dqkv = torch.empty((3, T, D), device=dev)
    dkr = torch.empty_like(kr)
    per_session = kr.shape[0] == B * 2 * L and B > 1
    part = torch.empty(max(1, lib.t4r_xlnet_attn_block_bwd_part_floats(B, L, D, n_head)), device=dev)
    rc = lib.t4r_xlnet_attn_block_bwd(_stream(), _chk(dy, torch.float32), _chk(saved["ao"]), _chk(h), _chk(saved["mean"]), _chk(saved["rstd"]),
         _chk(gamma), planes.data_ptr(), _chk(wq), _chk(wk), _chk(wv), _chk(saved["qkv"]), _chk(kr), 2 * L * D if per_session else 0,
         _chk(r_w_bias), _chk(r_r_bias), _chk(saved["lse"]), dh.data_ptr(), dao.data_ptr(), dqkv.data_ptr(), dkr.data_ptr(),
         _chk(d_rw), _chk(d_rr), _chk(d_gamma), _chk(d_beta), part.data_ptr(), B, L, D, n_head, float(drop_p), int(seed), int(ctr_prob),
         int(ctr_out), _p(key_len, torch.int32))
    if rc != 0:
        _lib.check(rc, "t4r_xlnet_attn_block_bwd")
    return dh, dao, dqkv, dkr


