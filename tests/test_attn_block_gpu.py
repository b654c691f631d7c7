"""The attention half of an XLNet layer as one kernel per direction (csrc/xlnet_attn_block.hip, exact fp32 matrix
instructions; the product library holds the FORWARD -- the one-kernel backward and the two-workgroups-per-CU forward layout
were measured slower and live, with their tests, in tools/experimental/) called through the C ABI, against an fp64 restatement of the chain it replaces -- HF modeling_xlnet.py
XLNetRelativeAttention.forward :245-282 (g = None), rel_attn_core :96-140 with rel_shift_bnij :86-94, post_attention
:142-152, reached through transformers4rec/torch/block/transformer.py:179-199 -- and its autograd:
  * d_model 32 / 64 / 128, d_head 16 / 32, L = 7 ... 32 (whole sessions per workgroup: 80 // L), batches that leave a
    ragged last workgroup, L not a multiple of 4 (Philox blocks straddle rows);
  * shared k_r (p = 0) and per-session k_r (p > 0), the opt-in padding mask;
  * training mode with the Philox masks exported by the same device function and fed to the reference;
  * against the four-launch form it replaces (same masks: identical dropout decisions).
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ORDER = ("q", "k", "v", "o", "r", "r_w_bias", "r_r_bias", "ln_w", "ln_b", "w1", "b1", "w2", "b2", "ff_ln_w", "ff_ln_b")
SEED, CTR_P, CTR_O = 1234, (5 << 16) | (1 << 8) | 2, (5 << 16) | (1 << 8) | 3


@pytest.fixture(scope="module")
def ops():
    from transformers4rec_amd import ops as o
    return o


def cu(t):
    return t.to(DEV).float().contiguous()


def _params(g, D, n, scale=0.15):
    dh = D // n
    r = lambda *s: scale * torch.randn(*s, generator=g, dtype=torch.float64)
    p = dict(q=r(D, n, dh), k=r(D, n, dh), v=r(D, n, dh), o=r(D, n, dh), r=r(D, n, dh), r_w_bias=r(n, dh), r_r_bias=r(n, dh),
             ln_w=1 + r(D), ln_b=r(D), w1=r(4 * D, D), b1=r(4 * D), w2=r(D, 4 * D), b2=r(D), ff_ln_w=1 + r(D), ff_ln_b=r(D))
    return {k: v.float().double() for k, v in p.items()}      # the values the kernels see


def _mask(ops, shape, p, seed, ctr):
    n = int(np.prod(shape))
    _, m = ops.dropout(torch.ones(1, device=DEV), p, seed, ctr, n_total=n, want_mask=True)
    return m.view(shape).double().cpu() / (1.0 - p)


def reference(p, h, kr, B, L, n, eps, mask_p, mask_o, key_len=None):
    """fp64, batch-first; kr [2L, D] or [B, 2L, D]; masks already scaled by 1 / (1 - p) (ones when p = 0)"""
    D = h.shape[1]
    dh = D // n
    hb = h.view(B, L, D)
    q = torch.einsum("bld,dnh->bnlh", hb, p["q"])
    k = torch.einsum("bld,dnh->bnlh", hb, p["k"])
    v = torch.einsum("bld,dnh->bnlh", hb, p["v"])
    krb = (kr if kr.dim() == 3 else kr.unsqueeze(0).expand(B, -1, -1)).reshape(B, 2 * L, n, dh).permute(0, 2, 1, 3)
    ac = torch.einsum("bnih,bnjh->bnij", q + p["r_w_bias"][None, :, None, :], k)
    raw = torch.einsum("bnih,bnmh->bnim", q + p["r_r_bias"][None, :, None, :], krb)
    i = torch.arange(L)[:, None]
    j = torch.arange(L)[None, :]
    bd = torch.gather(raw, 3, (j + L - i).expand(B, n, L, L))
    s = (ac + bd) / math.sqrt(dh)
    if key_len is not None:
        dead = (j[None] >= key_len.view(B, 1, 1)) & (j != i)[None]
        s = torch.where(dead[:, None], torch.full_like(s, -1e30), s)
    lse = torch.logsumexp(s, dim=-1)
    P = torch.softmax(s, dim=-1)
    vec = torch.einsum("bnij,bnjh->binh", P * mask_p, v).reshape(B * L, D)
    ao = vec @ p["o"].reshape(D, D).t()
    x = ao * mask_o + h
    mean = x.mean(1)
    rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False) + eps)
    h1 = (x - mean[:, None]) * rstd[:, None] * p["ln_w"] + p["ln_b"]
    qkv = torch.stack([t.permute(0, 2, 1, 3).reshape(B * L, D) for t in (q, k, v)])
    return dict(qkv=qkv, av=vec, lse=lse, ao=ao, mean=mean, rstd=rstd, h1=h1, _heads=(q, k, v))


def rel_err(a, ref):
    return float((a.double().cpu() - ref).abs().max() / ref.abs().max())


# (B, L, D, n_head, drop_p, key_len?)
CASES = [(9, 20, 128, 4, 0.0, False), (9, 20, 128, 4, 0.3, False), (5, 32, 64, 4, 0.0, False), (7, 13, 32, 2, 0.25, False),
         (3, 20, 64, 2, 0.3, True), (4, 7, 128, 8, 0.0, True), (33, 20, 128, 4, 0.3, True), (2, 16, 32, 1, 0.0, False)]


def _setup(ops, B, L, D, n, drop_p, with_len, seed=0):
    g = torch.Generator().manual_seed(100 * B + L + D + seed)
    p = _params(g, D, n)
    T = B * L
    h = torch.randn(T, D, generator=g, dtype=torch.float64).float().double()
    per_session = drop_p > 0
    kr = (0.3 * torch.randn(*((B, 2 * L, D) if per_session else (2 * L, D)), generator=g, dtype=torch.float64)).float().double()
    key_len = torch.randint(1, L + 1, (B,), generator=g) if with_len else None
    planes = ops.xlnet_layer_prepare([cu(p[k]) for k in ORDER], D)
    return p, h, kr, key_len, planes


@pytest.mark.parametrize("B,L,D,n,drop_p,with_len", CASES)
def test_attn_block_forward_matches_fp64(ops, B, L, D, n, drop_p, with_len):
    assert ops.xlnet_attn_block_supported(L, D, n)
    p, h, kr, key_len, planes = _setup(ops, B, L, D, n, drop_p, with_len)
    T = B * L
    kl = None if key_len is None else key_len.to(DEV).to(torch.int32)
    h1, saved = ops.xlnet_attn_block_fwd(cu(h), planes, cu(p["o"]).view(D, D), cu(kr).view(-1, D), cu(p["r_w_bias"]).view(-1),
                                          cu(p["r_r_bias"]).view(-1), cu(p["ln_w"]), cu(p["ln_b"]), B, L, n, 0.03, drop_p, SEED,
                                          CTR_P, CTR_O, key_len=kl)
    mp = _mask(ops, (B, n, L, L), drop_p, SEED, CTR_P) if drop_p > 0 else torch.ones(B, n, L, L, dtype=torch.float64)
    mo = _mask(ops, (T, D), drop_p, SEED, CTR_O) if drop_p > 0 else torch.ones(T, D, dtype=torch.float64)
    ref = reference(p, h, kr, B, L, n, 0.03, mp, mo, key_len)
    # (round 6: the q | k | v projections run on the two-way fp16 split -- 3-5e-6 of the largest output, like every other large
    #  contraction of the step; the attention core and the o-projection stay exact fp32)
    for name in ("qkv", "av", "ao", "h1", "mean", "rstd"):
        assert rel_err(saved[name] if name != "h1" else h1, ref[name]) < 6e-6, name
    assert float((saved["lse"].double().cpu() - ref["lse"]).abs().max()) < 6e-6 * max(1.0, float(ref["lse"].abs().max()))
    assert bool(torch.isfinite(h1).all())


def test_attn_block_forward_equals_the_four_launch_form(ops):
    """same inputs, same Philox keys: the one-launch block and projection -> core -> o-projection + LayerNorm agree to
    fp32 rounding (the dropout decisions are identical, so nothing but summation order differs)"""
    B, L, D, n, drop_p = 21, 20, 128, 4, 0.3
    p, h, kr, _, planes = _setup(ops, B, L, D, n, drop_p, False, seed=3)
    args = (cu(p["r_w_bias"]).view(-1), cu(p["r_r_bias"]).view(-1))
    h1, saved = ops.xlnet_attn_block_fwd(cu(h), planes, cu(p["o"]).view(D, D), cu(kr).view(-1, D), *args, cu(p["ln_w"]), cu(p["ln_b"]),
                                          B, L, n, 0.03, drop_p, SEED, CTR_P, CTR_O)
    qkv = ops.xlnet_qkv_proj(cu(h), planes)
    av, lse = ops.xlnet_attn_fwd(qkv[0], qkv[1], qkv[2], cu(kr).view(-1, D), *args, B, L, n, drop=(drop_p, SEED, CTR_P))
    h1c, ao, mean, rstd = ops.xlnet_oproj_ln(av, cu(h), planes, cu(p["ln_w"]), cu(p["ln_b"]), 0.03, drop=(drop_p, SEED, CTR_O))
    for a, b in ((saved["qkv"], qkv), (saved["av"], av), (saved["lse"], lse), (saved["ao"], ao), (h1, h1c)):
        assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))
    # a dropped probability is dropped in both: exact zeros of attn_vec contributions cannot be compared directly, but the
    # dropped OUTPUT entries can (x = ao * m + h, so where m = 0 the normalised row uses h alone)
    assert bool(((saved["ao"] != 0) == (ao != 0)).all())


def test_attn_block_unsupported_shapes_are_refused(ops):
    from transformers4rec_amd import _lib
    assert not ops.xlnet_attn_block_supported(33, 128, 4)        # L > 32
    assert not ops.xlnet_attn_block_supported(20, 128, 16)       # d_head 8
    assert not ops.xlnet_attn_block_supported(20, 256, 8)        # d_model beyond the token-tile kernels
    with pytest.raises(_lib.T4RHipError, match="unsupported shape"):
        z = torch.zeros(40 * 256, 256, device=DEV)
        ops.xlnet_attn_block_fwd(z, z, z, z, z, z, z, z, 256, 40, 8, 0.03)


# ------------------------------------------------------------------------------------------ backward
def _autograd_reference(p, h, kr, B, L, n, eps, mp, mo, key_len, dy):
    """fp64 autograd through the restatement above: gradients of sum(h1 * dy)"""
    leaves = {k: p[k].clone().requires_grad_(True) for k in ("q", "k", "v", "o", "r_w_bias", "r_r_bias", "ln_w", "ln_b")}
    hh = h.clone().requires_grad_(True)
    krr = kr.clone().requires_grad_(True)
    pp = dict(p)
    pp.update(leaves)
    out = reference(pp, hh, krr, B, L, n, eps, mp, mo, key_len)
    for t in out["_heads"] + (out["ao"],):
        t.retain_grad()
    (out["h1"] * dy).sum().backward()
    g = {k: v.grad for k, v in leaves.items()}
    D = h.shape[1]
    dqkv = torch.stack([t.grad.permute(0, 2, 1, 3).reshape(B * L, D) for t in out["_heads"]])
    g.update(h=hh.grad, kr=krr.grad, dqkv=dqkv, dao=out["ao"].grad)
    return g


@pytest.mark.parametrize("B,L,D,n,per_session,klen", [
    (5, 20, 128, 4, True, False), (1030, 20, 128, 4, True, False), (7, 20, 128, 4, False, False),
    (6, 20, 128, 8, True, True), (9, 13, 64, 4, True, True), (3, 32, 64, 2, False, True), (4, 9, 32, 2, True, False),
    (2, 5, 32, 1, False, False), (600, 8, 64, 4, False, False),
    # beyond one wave (csrc/xlnet_attn_long.hip), with and without the key mask, per-session and shared k_r
    (3, 100, 64, 4, True, True), (2, 90, 32, 2, False, True), (2, 129, 64, 1, True, False)])
def test_attention_core_backward_planes_path(ops, monkeypatch, B, L, D, n, per_session, klen):
    """t4r_xlnet_attn_bwd with q | k | v (and d q | d k | d v) as planes of one [3][T][D] buffer -- the layout the layer holds
    them in -- against autograd of the fp32 restatement of HF modeling_xlnet.py rel_attn_core :96-140 with the SAME Philox
    mask, and against the separate-tensor call.  (With the A/B variant library of tools/experimental/ and
    T4R_XLNET_ATTN_CORE16=1 the planes call takes the one-wave-per-head fp32-MFMA core; the product library has one core.)"""
    g = torch.Generator().manual_seed(B + 3 * L + D)
    dh = D // n
    p, seed, ctr = 0.3, 5, ops.dropout_ctr_hi(4, 2, ops.SITE_PROB)
    q, k, v = (torch.randn(B, L, n, dh, generator=g).requires_grad_() for _ in range(3))
    kr = torch.randn(*((B,) if per_session else ()), 2 * L, n, dh, generator=g).requires_grad_()
    rw, rr = (0.5 * torch.randn(n, dh, generator=g)).requires_grad_(), (0.5 * torch.randn(n, dh, generator=g)).requires_grad_()
    key_len = torch.randint(1, L + 1, (B,), generator=g, dtype=torch.int32) if klen else None
    m = _mask(ops, (B, n, L, L), p, seed, ctr)
    ac = torch.einsum("bind,bjnd->bnij", q + rw, k)
    bd_full = torch.einsum("bind,bpnd->bnip" if per_session else "bind,pnd->bnip", q + rr, kr)
    idx = torch.arange(L)[None, :] + L - torch.arange(L)[:, None]
    bd = torch.gather(bd_full, 3, idx[None, None].expand(B, n, L, L))
    sc = (ac + bd) / dh ** 0.5
    if klen:
        j = torch.arange(L)
        dead = (j[None, None, :] >= key_len.long()[:, None, None]) & (j[None, None, :] != j[None, :, None])
        sc = sc.masked_fill(dead[:, None], float("-inf"))
    prob = torch.softmax(sc, 3) * m.float()            # _mask: already scaled by 1 / (1 - p)
    ref = torch.einsum("bnij,bjnd->bind", prob, v)
    dout = torch.randn(B, L, n, dh, generator=g)
    ref.backward(dout)
    f2 = lambda t: t.detach().reshape(-1, D).to(DEV).contiguous()
    qkv = torch.stack([f2(q), f2(k), f2(v)])
    rwd, rrd = f2(rw).reshape(-1), f2(rr).reshape(-1)
    kl = key_len.to(DEV) if klen else None
    out, lse = ops.xlnet_attn_fwd(qkv[0], qkv[1], qkv[2], f2(kr), rwd, rrd, B, L, n, drop=(p, seed, ctr), key_len=kl)
    torch.testing.assert_close(out.cpu(), ref.detach().reshape(-1, D), atol=5e-5, rtol=1e-4)
    res = {}
    for name, (a, b, c) in dict(planes=(qkv[0], qkv[1], qkv[2]), separate=(qkv[0].clone(), qkv[1].clone(), qkv[2].clone())).items():
        drw, drr = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
        dq, dk, dv, dkr = ops.xlnet_attn_bwd(a, b, c, f2(kr), rwd, rrd, out, lse, f2(dout), drw, drr, B, L, n,
                                             drop=(p, seed, ctr), key_len=kl)
        res[name] = [t.cpu() for t in (dq, dk, dv, dkr, drw, drr)]
    want = [q.grad.reshape(-1, D), k.grad.reshape(-1, D), v.grad.reshape(-1, D), kr.grad.reshape(-1, D), rw.grad.reshape(-1), rr.grad.reshape(-1)]
    for name, got in res.items():
        for gt, wt, what in zip(got, want, ("dq", "dk", "dv", "dkr", "drw", "drr")):
            # sums over up to B L terms (biases, shared k_r): absolute error scales with the sum's magnitude
            tol = 2e-4 * max(1.0, float(wt.abs().max()))
            assert float((gt - wt).abs().max()) <= tol, (name, what, float((gt - wt).abs().max()), tol)
