"""Tests of the measured-and-not-kept kernels (README.md in this directory).  They need the A/B variant library:

    bash tools/experimental/build_variant.sh            # -> tools/bin/libt4r_hip_exp.so
    T4R_HIP_LIB=tools/bin/libt4r_hip_exp.so python -m pytest tools/experimental/test_experiments_gpu.py -q

Not collected by `pytest tests/` (the product library does not contain these kernels)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from test_attn_block_gpu import (CASES, CTR_O, CTR_P, DEV, SEED, _autograd_reference, _mask, _setup, cu, ops, reference,  # noqa: E402,F401
                                 rel_err)
import exp_ops  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not exp_ops.available(), reason="needs the variant library (T4R_HIP_LIB)")]


@pytest.mark.parametrize("B,L,D,n,drop_p,with_len", [c for c in CASES if c[2] == 128] + [(5, 32, 128, 4, 0.3, True), (11, 13, 128, 8, 0.25, False),
                                                                                      (1030, 20, 128, 4, 0.3, False)])
def test_attn_block_forward_two_workgroups_per_cu_layout(ops, monkeypatch, B, L, D, n, drop_p, with_len):
    """T4R_XLNET_ATTN_BLOCK2=1: the 40-row / 4-wave layout of the same kernel (two workgroups per CU) against fp64, and its
    saved tensors against the default layout's: q | k | v, attn_vec, lse and the o-projection are the same contraction chains
    element by element (bit-identical); the LayerNorm sums associate differently (1e-6)."""
    p, h, kr, key_len, planes = _setup(ops, B, L, D, n, drop_p, with_len)
    T = B * L
    kl = None if key_len is None else key_len.to(DEV).to(torch.int32)
    run = lambda: ops.xlnet_attn_block_fwd(cu(h), planes, cu(p["o"]).view(D, D), cu(kr).view(-1, D), cu(p["r_w_bias"]).view(-1),
                                           cu(p["r_r_bias"]).view(-1), cu(p["ln_w"]), cu(p["ln_b"]), B, L, n, 0.03, drop_p, SEED,
                                           CTR_P, CTR_O, key_len=kl)
    monkeypatch.setenv("T4R_XLNET_ATTN_BLOCK2", "0")
    h1a, sa = run()
    monkeypatch.setenv("T4R_XLNET_ATTN_BLOCK2", "1")
    h1b, sb = run()
    for name in ("qkv", "av", "lse", "ao"):
        assert torch.equal(sa[name], sb[name]), name
    for a, b in ((h1a, h1b), (sa["mean"], sb["mean"]), (sa["rstd"], sb["rstd"])):
        assert float((a - b).abs().max()) < 2e-6 * max(1.0, float(a.abs().max()))
    if B <= 64:
        mp = _mask(ops, (B, n, L, L), drop_p, SEED, CTR_P) if drop_p > 0 else torch.ones(B, n, L, L, dtype=torch.float64)
        mo = _mask(ops, (T, D), drop_p, SEED, CTR_O) if drop_p > 0 else torch.ones(T, D, dtype=torch.float64)
        ref = reference(p, h, kr, B, L, n, 0.03, mp, mo, key_len)
        for name in ("qkv", "av", "ao", "h1", "mean", "rstd"):
            assert rel_err(sb[name] if name != "h1" else h1b, ref[name]) < 3e-6, name



@pytest.mark.parametrize("B,L,D,n,drop_p,with_len", CASES)
def test_attn_block_backward_matches_fp64_autograd(ops, B, L, D, n, drop_p, with_len):
    p, h, kr, key_len, planes = _setup(ops, B, L, D, n, drop_p, with_len, seed=1)
    T = B * L
    g = torch.Generator().manual_seed(7 + B)
    dy = torch.randn(T, D, generator=g, dtype=torch.float64).float().double()
    kl = None if key_len is None else key_len.to(DEV).to(torch.int32)
    mp = _mask(ops, (B, n, L, L), drop_p, SEED, CTR_P) if drop_p > 0 else torch.ones(B, n, L, L, dtype=torch.float64)
    mo = _mask(ops, (T, D), drop_p, SEED, CTR_O) if drop_p > 0 else torch.ones(T, D, dtype=torch.float64)
    ref = _autograd_reference(p, h, kr, B, L, n, 0.03, mp, mo, key_len, dy)
    rw, rr = cu(p["r_w_bias"]).view(-1), cu(p["r_r_bias"]).view(-1)
    krd = cu(kr).view(-1, D)
    h1, saved = ops.xlnet_attn_block_fwd(cu(h), planes, cu(p["o"]).view(D, D), krd, rw, rr, cu(p["ln_w"]), cu(p["ln_b"]), B, L, n, 0.03,
                                          drop_p, SEED, CTR_P, CTR_O, key_len=kl)
    # accumulated outputs start from a known non-zero state
    base = {k: torch.full((D,), 0.25, device=DEV) for k in ("rw", "rr", "gamma", "beta")}
    acc = {k: v.clone() for k, v in base.items()}
    dh, dao, dqkv, dkr = exp_ops.xlnet_attn_block_bwd(cu(dy), saved, cu(h), planes, cu(p["q"]).view(D, D), cu(p["k"]).view(D, D),
                                                  cu(p["v"]).view(D, D), krd, rw, rr, cu(p["ln_w"]), acc["rw"], acc["rr"],
                                                  acc["gamma"], acc["beta"], B, L, n, drop_p, SEED, CTR_P, CTR_O, key_len=kl)
    tol = 5e-6
    assert rel_err(dao, ref["dao"]) < tol
    assert rel_err(dqkv, ref["dqkv"]) < tol
    assert rel_err(dh, ref["h"]) < tol
    assert rel_err(dkr.view(ref["kr"].shape), ref["kr"]) < tol
    assert rel_err(acc["rw"] - 0.25, ref["r_w_bias"].reshape(-1)) < tol
    assert rel_err(acc["rr"] - 0.25, ref["r_r_bias"].reshape(-1)) < tol
    assert rel_err(acc["gamma"] - 0.25, ref["ln_w"]) < tol
    assert rel_err(acc["beta"] - 0.25, ref["ln_b"]) < tol
    # the weight gradients the caller forms from these rows
    hq = h.t() @ ref["dqkv"][0]
    assert rel_err((cu(h).double().t() @ dqkv[0].double()).view(D, n, D // n), ref["q"]) < 2e-5 and float(hq.abs().max()) > 0
    assert rel_err((dao.double().t() @ saved["av"].double()).view(D, n, D // n), ref["o"]) < 2e-5
    # no atomics: a second call gives the same bits
    acc2 = {k: v.clone() for k, v in base.items()}
    dh2, dao2, dqkv2, dkr2 = exp_ops.xlnet_attn_block_bwd(cu(dy), saved, cu(h), planes, cu(p["q"]).view(D, D), cu(p["k"]).view(D, D),
                                                      cu(p["v"]).view(D, D), krd, rw, rr, cu(p["ln_w"]), acc2["rw"], acc2["rr"],
                                                      acc2["gamma"], acc2["beta"], B, L, n, drop_p, SEED, CTR_P, CTR_O, key_len=kl)
    assert torch.equal(dh, dh2) and torch.equal(dqkv, dqkv2) and torch.equal(dkr, dkr2)
    assert all(torch.equal(acc[k], acc2[k]) for k in acc)


