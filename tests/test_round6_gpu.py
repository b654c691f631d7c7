"""Round 6 (GPU): parity AT THE BENCHMARKED SETTING.

  * the oracle's integer restatement of the device random streams (oracle/device_rng.py) against masks exported from
    the device, bit for bit, every dropout site and the MLM draws;
  * one whole training step of BASELINE configs[1] and configs[2] at FULL size (B 1024, V 100 001 rows, d 128, 4 layers)
    with dropout 0.3 -- the configuration bench.py times -- against the CPU oracle given the SAME decisions at every
    site (input, the pos_emb mask shared by the four layers, prob / attn_out / ff_act / ff_out per layer, output):
    loss, logits (the north_star 1e-3 gate, asserted much tighter) and the item-table / layer / projection gradients;
  * three optimizer steps of that configuration in lockstep (Adam both sides, masks and MLM targets from the
    restatement alone -- nothing is read back from the device but the results).
"""
import os
import sys

import numpy as np
import pytest
import torch

import device_rng as R
import golden_utils as gu
import t4r_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ops():
    import build_c
    from transformers4rec_amd import ops as _ops

    build_c.build()
    return _ops


def close(a, b, rtol=1e-4, atol=1e-5, msg=None):
    torch.testing.assert_close(a.detach().cpu().float(), b.detach().cpu().float(), rtol=rtol, atol=atol, msg=msg)


@pytest.mark.parametrize("n", [1, 5, 4096, 100_003])
@pytest.mark.parametrize("p", [0.1, 0.3])
def test_dropout_site_restatement_equals_the_device_mask(ops, n, p):
    seed = 0x1234_5678_9ABC_DEF1 & 0x7FFFFFFFFFFFFFFF
    for offset, layer, site in ((1, 0, ops.SITE_PROB), (70_000, 3, ops.SITE_FF_ACT), (2, 255, ops.SITE_POS),
                                (1, 255, ops.SITE_INPUT), (9, 255, ops.SITE_FINAL), (5, 1, ops.SITE_ATTN_OUT), (5, 1, ops.SITE_FF_OUT)):
        ctr = ops.dropout_ctr_hi(offset, layer, site)
        assert ctr == R.dropout_ctr_hi(offset, layer, site)
        _, m = ops.dropout(torch.ones(1, device=DEV), p, seed, ctr, n_total=n, want_mask=True)
        assert np.array_equal(m.cpu().numpy(), R.dropout_keep(seed, ctr, n, p)), (offset, layer, site)


@pytest.mark.parametrize("B,L", [(1024, 20), (257, 33), (64, 64), (33, 255), (21, 300), (9, 1023)])
def test_mlm_draw_restatement_equals_the_device_targets(ops, B, L):
    g = torch.Generator().manual_seed(B + L)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[:8] = torch.tensor([1, 1, 2, 2, L, L, 3, 1])
    ids = torch.randint(1, 100_000, (B, L), generator=g) * (torch.arange(L)[None] < lens[:, None])
    for seed, offset in ((1234, 0), (4321, 7 * B * L), (2**62 + 11, 2**33 + 5)):
        mask, labels, counts = ops.mask_targets(ids.to(DEV), ops.MLM_TRAIN, 0, None, None, None, 0.15, seed, offset)
        m_ref, lab_ref = R.mlm_targets_train_device(ids, seed, offset, 0.15)
        assert torch.equal(mask.cpu(), m_ref) and torch.equal(labels.cpu(), lab_ref)
        assert torch.equal(counts.cpu().long(), m_ref.sum(1))


def _bench_model(tr, multi, dropout, V=100_000, D=128, L=20, n_layer=4, lr=None):
    cats = (("category", 1000), ("brand", 100), ("kind", 10)) if multi else ()
    conts = ("price", "age") if multi else ()
    schema = tr.session_schema(V, L, cats, conts)
    kw = dict(max_sequence_length=L, masking="mlm")
    if multi:
        kw.update(continuous_soft_embeddings=True, d_output=D, embedding_dims={"item_id": D}, embedding_dim_default=64)
    else:
        kw.update(embedding_dim_default=D)
    torch.manual_seed(0)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, **kw)
    cfg = tr.XLNetConfig.build(D, 4, n_layer, total_seq_length=L, dropout=dropout)
    model = cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=True))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(DEV).train()
    return schema, model, sd


@pytest.mark.parametrize("multi", [False, True], ids=["configs1", "configs2"])
def test_full_size_dropout_step_vs_oracle(ops, multi):
    """the benchmarked configuration, dropout 0.3, one training step: HIP vs the CPU oracle taking the same decisions."""
    import transformers4rec_amd as tr

    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    B, L, V, D, NL, p_drop = 1024, 20, 100_000, 128, 4, 0.3
    schema, model, sd = _bench_model(tr, multi, p_drop)
    xl, masking = model.transformer_block.transformer, model.input_features.masking
    masking.seed, xl.seed = 1234, 4321          # bench.rank_seeds(0)
    data = tr.random_data_from_schema(schema, B, L, seed=11)
    out = model({k: v.to(DEV) for k, v in data.items()}, training=True)
    out["loss"].backward()
    assert xl._drop_offset == 1
    # the decisions, from the restatement alone
    mask, labels = R.mlm_targets_train_device(data["item_id"], 1234, 0, 0.15)
    assert torch.equal(masking.mask_schema.cpu(), mask) and torch.equal(masking.masked_targets.cpu(), labels)
    masks = R.xlnet_dropout_masks(B, L, D, 4, NL, p_drop, seed=4321, offset=1)
    p = gu.oracle_params({"p/" + k: v.numpy() for k, v in sd.items()}, requires_grad=True)
    ref = O.session_forward(p, dict(n_head=4, eps=0.03, item="item_id", masking="mlm"), data, mask, labels, True, False,
                            drop=(p_drop, masks))
    ref["loss"].backward()
    dl = abs(float(out["loss"].detach()) - float(ref["loss"].detach()))
    dp = float((out["predictions"].detach().cpu() - ref["logits"].detach()).abs().max())
    print(f"\n[dropout {p_drop} full size, multi={multi}] loss {float(ref['loss']):.6f}: |d loss| {dl:.2e}, max |d logits| {dp:.2e} "
          f"over {tuple(ref['logits'].shape)}")
    assert torch.equal(out["labels"].cpu(), ref["labels"])
    assert dl < 1e-4 and dp < 1e-4                       # north_star: 1e-3
    assert abs(float(ref["loss"]) - np.log(V + 1)) < 0.35   # ~ln(V) at init (dropout raises it a little)
    # the masks mattered: the dropout-0 forward of the same model is elsewhere
    with torch.no_grad():
        ref0 = O.session_forward(p, dict(n_head=4, eps=0.03, item="item_id", masking="mlm"), data, mask, labels, True, False)
    assert float((ref0["logits"] - ref["logits"].detach()).abs().max()) > 10 * dp
    g_hip = model.input_features.item_embedding_table.weight.grad.cpu()
    g_ref = p["tables"]["item_id"].grad
    close(g_hip, g_ref, rtol=1e-3, atol=2e-7)
    close(g_hip.sum(0), g_ref.sum(0), rtol=1e-3, atol=1e-6)
    close(masking.masked_item_embedding.grad, p["masked_item_embedding"].grad, rtol=2e-3, atol=2e-7)
    for i in (0, NL - 1):
        lay, lp = xl.layer[i], p["layers"][i]
        close(lay.ff.layer_1.weight.grad, lp["w1"].grad, rtol=2e-3, atol=3e-7, msg=lambda m, i=i: f"layer {i} w1: {m}")
        close(lay.ff.layer_2.weight.grad, lp["w2"].grad, rtol=2e-3, atol=3e-7, msg=lambda m, i=i: f"layer {i} w2: {m}")
        close(lay.rel_attn.q.grad, lp["q"].grad, rtol=2e-3, atol=3e-7, msg=lambda m, i=i: f"layer {i} q: {m}")
        close(lay.rel_attn.o.grad, lp["o"].grad, rtol=2e-3, atol=3e-7, msg=lambda m, i=i: f"layer {i} o: {m}")
        close(lay.rel_attn.r.grad, lp["r"].grad, rtol=2e-3, atol=3e-7, msg=lambda m, i=i: f"layer {i} r: {m}")
        close(lay.ff.layer_norm.weight.grad, lp["ff_ln_w"].grad, rtol=2e-3, atol=3e-6, msg=lambda m, i=i: f"layer {i} ln2: {m}")
    if multi:
        close(model.input_features.projection_module[0][0].weight.grad, p["proj"][0].grad, rtol=2e-3, atol=3e-7)
        close(model.input_features.continuous_module.embedding_tables["price"].embedding_table.weight.grad,
              p["soft"]["price"][2].grad, rtol=2e-3, atol=2e-6)


@pytest.mark.parametrize("L,D", [(100, 64), (70, 128), (20, 256), (24, 192), (100, 256)])
def test_long_sequence_dropout_step_vs_oracle(ops, L, D):
    """total_seq_length beyond one wave of the attention kernels, or a d_model / head width beyond the fused kernels (d_model 256 and
    192 with four heads: d_head 64 and 48) -- csrc/xlnet_attn_long.hip --, training mode, dropout 0.3: one step
    against the CPU oracle taking the same decisions (MLM targets and every dropout site from the restatement alone).  The reference
    takes any length (config/transformer.py:432-482)."""
    import transformers4rec_amd as tr

    B, V, NL, p_drop = 24, 3000, 2, 0.3
    schema, model, sd = _bench_model(tr, False, p_drop, V=V, D=D, L=L, n_layer=NL)
    xl, masking = model.transformer_block.transformer, model.input_features.masking
    masking.seed, xl.seed = 77, 78
    data = tr.random_data_from_schema(schema, B, L, seed=5)
    out = model({k: v.to(DEV) for k, v in data.items()}, training=True)
    out["loss"].backward()
    mask, labels = R.mlm_targets_train_device(data["item_id"], 77, 0, 0.15)
    assert torch.equal(masking.mask_schema.cpu(), mask) and torch.equal(masking.masked_targets.cpu(), labels)
    masks = R.xlnet_dropout_masks(B, L, D, 4, NL, p_drop, seed=78, offset=1)
    p = gu.oracle_params({"p/" + k: v.numpy() for k, v in sd.items()}, requires_grad=True)
    ref = O.session_forward(p, dict(n_head=4, eps=0.03, item="item_id", masking="mlm"), data, mask, labels, True, False,
                            drop=(p_drop, masks))
    ref["loss"].backward()
    assert torch.equal(out["labels"].cpu(), ref["labels"])
    assert abs(float(out["loss"].detach()) - float(ref["loss"].detach())) < 1e-4
    assert float((out["predictions"].detach().cpu() - ref["logits"].detach()).abs().max()) < 1e-4
    close(model.input_features.item_embedding_table.weight.grad, p["tables"]["item_id"].grad, rtol=1e-3, atol=1e-6)
    for i in range(NL):
        lay, lp = xl.layer[i], p["layers"][i]
        for name, got in (("q", lay.rel_attn.q.grad), ("k", lay.rel_attn.k.grad), ("v", lay.rel_attn.v.grad), ("o", lay.rel_attn.o.grad),
                          ("r", lay.rel_attn.r.grad), ("r_w_bias", lay.rel_attn.r_w_bias.grad), ("r_r_bias", lay.rel_attn.r_r_bias.grad),
                          ("w1", lay.ff.layer_1.weight.grad), ("w2", lay.ff.layer_2.weight.grad)):
            close(got, lp[name].grad, rtol=2e-3, atol=2e-6, msg=lambda m, i=i, name=name: f"layer {i} {name}: {m}")


def test_three_optimizer_steps_in_lockstep_at_the_benchmarked_setting(ops):
    """configs[1], dropout 0.3, Adam lr 1e-3 (bench.py's optimizer), three steps: every step's loss and the parameters after
    the third against the CPU oracle driven by the restated streams only (MLM offset advances by B*L per step, the dropout
    offset by one per forward)."""
    import transformers4rec_amd as tr

    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    B, L, V, D, NL, p_drop, lr = 1024, 20, 100_000, 128, 4, 0.3, 1e-3
    schema, model, sd = _bench_model(tr, False, p_drop)
    xl, masking = model.transformer_block.transformer, model.input_features.masking
    masking.seed, xl.seed = 1234, 4321
    dense, tables = tr.flatten_model(model)
    opt = tr.FusedAdam([dense, tables], lr=lr)
    p = gu.oracle_params({"p/" + k: v.numpy() for k, v in sd.items()}, requires_grad=True)
    leaves = [p["tables"]["item_id"], p["masked_item_embedding"]] + [t for lp in p["layers"] for t in lp.values()]
    ref_opt = torch.optim.Adam(leaves, lr=lr)
    cfg_o = dict(n_head=4, eps=0.03, item="item_id", masking="mlm")
    for step in range(3):
        ids = tr.random_data_from_schema(schema, B, L, seed=100 + step)["item_id"]
        out = model({"item_id": ids.to(DEV)}, training=True)
        out["loss"].backward()
        opt.step()
        mask, labels = R.mlm_targets_train_device(ids, 1234, step * B * L, 0.15)
        masks = R.xlnet_dropout_masks(B, L, D, 4, NL, p_drop, seed=4321, offset=step + 1)
        ref_opt.zero_grad()
        ref = O.session_forward(p, cfg_o, {"item_id": ids}, mask, labels, True, False, drop=(p_drop, masks))
        ref["loss"].backward()
        ref_opt.step()
        assert torch.equal(masking.masked_targets.cpu(), labels)
        dl = abs(float(out["loss"].detach()) - float(ref["loss"].detach()))
        print(f"\n[lockstep step {step}] loss {float(ref['loss']):.6f} |d| {dl:.2e}")
        assert dl < 1e-4
    # Adam's first steps move every touched weight by ~lr whatever the gradient's size (the update is g / |g| at step 1), so
    # an element whose gradient is rounding noise may move the other way: bound the FRACTION of such elements, and everything
    # else at a small fraction of lr
    def adam_close(a, b, name):
        d = (a.detach().cpu() - b.detach()).abs()
        frac = float((d > 0.05 * lr).float().mean())
        print(f"[lockstep] {name}: max |d| {float(d.max()):.2e} ({float(d.max()) / lr:.2f} lr), fraction beyond 0.05 lr: {frac:.2e}")
        assert frac < 1e-3 and float(d.max()) <= 6.5 * lr, name

    adam_close(model.input_features.item_embedding_table.weight, p["tables"]["item_id"], "item table")
    adam_close(xl.layer[1].ff.layer_1.weight, p["layers"][1]["w1"], "layer 1 w1")
    adam_close(xl.layer[3].rel_attn.q, p["layers"][3]["q"], "layer 3 q")


# ------------------------------------------------------------------------------------------ small launches folded (round 6)
@pytest.mark.parametrize("T,D,n", [(20480, 128, 2780), (37, 64, 5), (64, 32, 0), (100, 128, 100)])
def test_scatter_rows_dense_equals_zero_fill_plus_scale_plus_scatter(ops, T, D, n):
    """one launch for the backward of the label-row selection (prediction_task.py:472-479): bit-identical to torch.zeros +
    multiply + scatter_rows_add_"""
    g = torch.Generator().manual_seed(T + n)
    pos = torch.sort(torch.randperm(T, generator=g)[:n])[0].to(torch.int32)
    pos_dev = torch.zeros(T, dtype=torch.int32)
    pos_dev[:n] = pos
    src = torch.randn(max(n, 1), D, generator=g).to(DEV)
    scale = torch.tensor(0.37, device=DEV)
    got = ops.scatter_rows_dense(src, pos_dev.to(DEV), n, scale, T)
    want = torch.zeros(T, D, device=DEV)
    if n:
        ops.scatter_rows_add_((src[:n] * scale).contiguous(), pos_dev[:n].to(DEV), want)
    assert torch.equal(got, want)
    # no scale pointer == a scale of one
    assert torch.equal(ops.scatter_rows_dense(src, pos_dev.to(DEV), n, None, T),
                       ops.scatter_rows_dense(src, pos_dev.to(DEV), n, torch.tensor(1.0, device=DEV), T))


@pytest.mark.parametrize("H", [128, 64, 96, 30])
@pytest.mark.parametrize("mode", ["MASK_MLM", "MASK_CLM", "MASK_CLM_INFER"])
def test_out_of_place_mask_backward(ops, mode, H):
    """dx / d masked_item_embedding of the mask backward (reference: autograd of masking.py:302-337 / :473-498): the out-of-place
    entry leaves the incoming gradient untouched, equals the in-place one bit for bit, and both equal the definition
    (H % 4 == 0: the float4 kernel; H = 30: the scalar one)"""
    B, L = 33, 20
    g = torch.Generator().manual_seed(3)
    dy = torch.randn(B, L, H, generator=g).to(DEV)
    mask = (torch.rand(B, L, generator=g) < 0.3).to(DEV)
    keep = dy.clone()
    m1, m2 = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    dx = ops.apply_mask_bwd(dy, mask, m1, getattr(ops, mode))
    assert torch.equal(dy, keep)                                  # the incoming gradient is untouched
    ref = dy.clone()
    ops.apply_mask_bwd_(ref, mask, m2, getattr(ops, mode))
    assert torch.equal(dx, ref) and torch.equal(m1, m2)
    # the definition: MLM replaces the masked positions, CLM (training) the UNmasked ones and zeroes the last position,
    # CLM inference the unmasked ones
    replaced = mask if mode == "MASK_MLM" else ~mask
    want = torch.where(replaced[..., None], torch.zeros_like(dy), dy)
    if mode == "MASK_CLM":
        want[:, L - 1] = 0
    assert torch.equal(dx, want)
    torch.testing.assert_close(m1, (dy * replaced[..., None]).double().sum((0, 1)).float(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,L,D,n", [(9, 20, 128, 4), (300, 20, 64, 4)])
def test_output_dropout_fused_into_the_last_layer(ops, B, L, D, n):
    """layer_idx | LAYER_FUSE_FINAL: the model-level output dropout (HF modeling_xlnet.py:1177) applied by the last layer's
    feed-forward kernels == the layer without the flag followed by the element-wise dropout of that site; backward likewise
    (d h and every parameter gradient), bit for bit in the per-row results"""
    import test_kernels_gpu as K

    g = torch.Generator().manual_seed(B + D)
    prm = K._layer_params(g, D, n)
    params = [K.cu(prm[k]) for k in K.ORDER]
    h = K.cu(torch.randn(B * L, D, generator=g))
    dout = K.cu(torch.randn(B * L, D, generator=g))
    pos = K.cu(O.xlnet_pos_emb(L, D))
    p_drop, seed, offset, layer = 0.3, 1234, 5, 3
    kw = dict(drop_p=p_drop, seed=seed, offset=offset)
    ctr_final = ops.dropout_ctr_hi(offset, 255, ops.SITE_FINAL)
    # reference composition: plain layer, then the element-wise dropout; backward: dropout of the gradient, then the layer
    out0, ws0 = ops.xlnet_layer_fwd(h, pos, params, B, L, n, 0.03, layer_idx=layer, **kw)
    want = ops.dropout(out0.view(-1), p_drop, seed, ctr_final).view(B * L, D)
    g0 = [torch.zeros_like(t) for t in params]
    dh0 = ops.xlnet_layer_bwd(h, pos, params, g0, ws0, ops.dropout(dout.view(-1), p_drop, seed, ctr_final).view(B * L, D),
                              B, L, n, 0.03, layer_idx=layer, **kw)
    out1, ws1 = ops.xlnet_layer_fwd(h, pos, params, B, L, n, 0.03, layer_idx=layer | ops.LAYER_FUSE_FINAL, **kw)
    assert torch.equal(out1, want)
    assert float((out1 == 0).float().mean()) > 0.25                 # the site is active
    g1 = [torch.zeros_like(t) for t in params]
    dh1 = ops.xlnet_layer_bwd(h, pos, params, g1, ws1, dout, B, L, n, 0.03, layer_idx=layer | ops.LAYER_FUSE_FINAL, **kw)
    torch.cuda.synchronize()
    assert torch.equal(dh1, dh0)
    for name, a, b in zip(K.ORDER, g1, g0):
        assert torch.equal(a, b), name


@pytest.mark.parametrize("n,rows", [(20480, 100_001), (1, 3), (5, 10), (63, 7), (64, 100), (1025, 262_143), (32768, 65_535),
                                    (4097, 1_000_001), (40_000, 100_001)])
def test_sort_ids_is_the_stable_order(ops, n, rows):
    """t4r_sort_ids gives THE stable order the deterministic table gradient is defined by: keys ascending, equal keys in ascending
    lookup order; padding and out-of-range ids carry `rows` and come last.  (Written for round 6's one-launch LDS sort, which
    passed it and was removed for being slower than the library's launches; kept as the sort's definition.)"""
    g = torch.Generator().manual_seed(n + rows)
    ids = torch.randint(0, rows, (n,), generator=g)
    if n > 8:
        ids[torch.randint(0, n, (max(1, n // 50),), generator=g)] = 0                  # padding
        ids[torch.randint(0, n, (max(1, n // 100),), generator=g)] = rows + 5           # out of range
        ids[torch.randint(0, n, (max(1, n // 100),), generator=g)] = -3
        ids[: n // 4] = ids[n // 4: 2 * (n // 4)]                                       # many duplicates
    keys, perm = ops.sort_ids(ids.to(DEV), rows, 0)
    key_ref = torch.where((ids == 0) | (ids < 0) | (ids >= rows), torch.full_like(ids, rows), ids)
    ks, ps = torch.sort(key_ref, stable=True)
    assert torch.equal(keys.cpu().long(), ks) and torch.equal(perm.cpu().long(), ps)
    # no padding id (the sampled head's rows): id 0 is a row like any other
    keys2, perm2 = ops.sort_ids(ids.to(DEV), rows, -1)
    key_ref2 = torch.where((ids < 0) | (ids >= rows), torch.full_like(ids, rows), ids)
    ks2, ps2 = torch.sort(key_ref2, stable=True)
    assert torch.equal(keys2.cpu().long(), ks2) and torch.equal(perm2.cpu().long(), ps2)


@pytest.mark.parametrize("B,L,D,n", [(9, 20, 128, 4), (300, 20, 64, 4), (5, 32, 128, 4)])
def test_input_dropout_fused_into_the_first_layer(ops, B, L, D, n):
    """layer_idx | LAYER_FUSE_INPUT: the model-level input dropout (HF modeling_xlnet.py:1116) applied by the first layer's
    attention-block kernel on load == the element-wise dropout of that site followed by the layer without the flag; backward:
    the same parameter gradients, and d h == the plain layer's d h masked again"""
    import test_kernels_gpu as K

    assert ops.xlnet_attn_block_supported(L, D, n)
    g = torch.Generator().manual_seed(B + D + L)
    prm = K._layer_params(g, D, n)
    params = [K.cu(prm[k]) for k in K.ORDER]
    h = K.cu(torch.randn(B * L, D, generator=g))
    dout = K.cu(torch.randn(B * L, D, generator=g))
    pos = K.cu(O.xlnet_pos_emb(L, D))
    p_drop, seed, offset = 0.3, 4321, 7
    kw = dict(drop_p=p_drop, seed=seed, offset=offset)
    ctr_in = ops.dropout_ctr_hi(offset, 255, ops.SITE_INPUT)
    hd = ops.dropout(h.view(-1), p_drop, seed, ctr_in).view(B * L, D)
    out0, ws0 = ops.xlnet_layer_fwd(hd, pos, params, B, L, n, 0.03, layer_idx=0, **kw)
    g0 = [torch.zeros_like(t) for t in params]
    dh0 = ops.xlnet_layer_bwd(hd, pos, params, g0, ws0, dout, B, L, n, 0.03, layer_idx=0, **kw)
    want_dh = ops.dropout(dh0.view(-1), p_drop, seed, ctr_in).view(B * L, D)
    out1, ws1 = ops.xlnet_layer_fwd(h, pos, params, B, L, n, 0.03, layer_idx=ops.LAYER_FUSE_INPUT, **kw)
    assert torch.equal(out1, out0)
    g1 = [torch.zeros_like(t) for t in params]
    dh1 = ops.xlnet_layer_bwd(h, pos, params, g1, ws1, dout, B, L, n, 0.03, layer_idx=ops.LAYER_FUSE_INPUT, **kw)
    torch.cuda.synchronize()
    assert torch.equal(dh1, want_dh)
    assert float((dh1 == 0).float().mean()) > 0.25
    for name, a, b in zip(K.ORDER, g1, g0):
        assert torch.equal(a, b), name
    # both flags on one layer (a one-layer stack)
    both = ops.LAYER_FUSE_INPUT | ops.LAYER_FUSE_FINAL
    out2, ws2 = ops.xlnet_layer_fwd(h, pos, params, B, L, n, 0.03, layer_idx=both, **kw)
    ctr_fin = ops.dropout_ctr_hi(offset, 255, ops.SITE_FINAL)
    assert torch.equal(out2, ops.dropout(out0.view(-1), p_drop, seed, ctr_fin).view(B * L, D))


def test_table_maximum_from_the_optimizer_launch(ops):
    """FusedAdam leaves the per-workgroup maxima of the tied item table in its launch (t4r_adam_step_amax); the next step's head
    takes max |W| from them (t4r_head_split_w_amax_hint) instead of a memset + a pass over the table: same bits in logits, loss
    and d X; the hint is dropped as soon as anything else writes the table."""
    import transformers4rec_amd as tr

    B, L, V, D = 256, 20, 30_000, 128
    schema = tr.session_schema(V, L)
    torch.manual_seed(0)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="mlm", embedding_dim_default=D)
    model = tr.XLNetConfig.build(D, 4, 2, total_seq_length=L, dropout=0.0).to_torch_model(
        inputs, tr.NextItemPredictionTask(weight_tying=True)).to(DEV).train()
    dense, tables = tr.flatten_model(model)
    opt = tr.FusedAdam([dense, tables], lr=1e-2)
    W = model.input_features.item_embedding_table.weight
    assert ops.w_amax_of(W) is None
    x = {"item_id": tr.random_data_from_schema(schema, B, L, seed=3)["item_id"].to(DEV)}
    model(x, training=True)["loss"].backward()
    opt.step()
    rec = ops.w_amax_of(W)
    assert rec is not None
    part, n = rec
    assert 1 <= n <= 512 and float(part[:n].max()) == float(W.detach().abs().max())
    # the head with and without the hint: bit-identical
    N = 700
    g = torch.Generator(device=DEV).manual_seed(1)
    xr = torch.randn(N, D, device=DEV, generator=g)
    lab = torch.randint(1, V + 1, (N,), device=DEV, generator=g)
    outs = []
    for hint in (None, rec):
        ws = ops.head_split_prepare(xr, V + 1)
        outs.append(ops.head_split_logits_ce_dx(ws, xr, W.detach(), lab, ldc=ops.pad_ld(V + 1), w_amax=hint))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    # a training step that uses the hint == the same step without it (fresh model, same seeds)
    def two_steps(use_hint):
        torch.manual_seed(0)
        inp = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="mlm", embedding_dim_default=D)
        mdl = tr.XLNetConfig.build(D, 4, 2, total_seq_length=L, dropout=0.0).to_torch_model(
            inp, tr.NextItemPredictionTask(weight_tying=True)).to(DEV).train()
        mdl.input_features.masking.seed = 5
        dn, tb = tr.flatten_model(mdl)
        o = tr.FusedAdam([dn, tb], lr=1e-2)
        if not use_hint:
            o._amax_targets = [None for _ in o._amax_targets]
        losses = []
        for _ in range(3):
            out = mdl(x, training=True)
            out["loss"].backward()
            o.step()
            losses.append(float(out["loss"].detach()))
        return losses, tb.data.clone()

    la, ta = two_steps(True)
    lb, tb_ = two_steps(False)
    # (at this small size a few reductions of the body still use fp32 atomics: two runs of the SAME setting differ in the last
    # bits of the parameters too; the head itself is compared bit for bit above)
    assert max(abs(a - b) for a, b in zip(la, lb)) < 1e-5 and float((ta - tb_).abs().max()) < 1e-6
    # anything else writing the table invalidates the hint
    with torch.no_grad():
        W.mul_(1.0)
    assert ops.w_amax_of(W) is None


@pytest.mark.parametrize("switch", ["_GEN_POS", "_FUSE_INPUT", "_FUSE_FINAL"])
def test_model_level_dropout_fusions_do_not_change_the_step(ops, switch, monkeypatch):
    """the three model-level dropout sites of HF XLNetModel (input :1116, pos_emb :1143, output :1177) ride inside the stack's own
    kernels (first layer, stack prologue, last layer); with each fusion switched off the element-wise launches run instead: same
    hidden states, same gradients, bit for bit"""
    import transformers4rec_amd as tr
    from transformers4rec_amd import transformer as T

    B, L, D = 96, 20, 128
    torch.manual_seed(0)
    xl = tr.XLNetConfig.build(D, 4, 3, total_seq_length=L, dropout=0.3).to_huggingface_torch_model().to(DEV).train()
    xl.seed = 99
    x = torch.randn(B, L, D, device=DEV)
    dout = torch.randn(B, L, D, device=DEV)

    def run(on):
        monkeypatch.setattr(T, switch, on)
        xl._drop_offset = 0
        for p in xl.parameters():
            p.grad = None
        xin = x.clone().requires_grad_()
        (h,) = xl(inputs_embeds=xin)
        h.backward(dout)
        torch.cuda.synchronize()
        return h.detach().clone(), xin.grad.clone(), {n: p.grad.clone() for n, p in xl.named_parameters() if p.grad is not None}

    h1, g1, p1 = run(True)
    h0, g0, p0 = run(False)
    assert float((h1 == 0).float().mean()) > 0.25 and float((g1 == 0).float().mean()) > 0.25
    assert torch.equal(h1, h0) and torch.equal(g1, g0)
    assert p1.keys() == p0.keys() and len(p1) >= 3 * 13
    for n in p1:
        assert torch.equal(p1[n], p0[n]), n
