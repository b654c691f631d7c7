"""Round 5 (GPU): the fallback of the recomputing head, the step next to a co-resident "collective" (CU occupier), the
measurement kernels of tools/t4r_tools.hip."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _model(tr, V, D, L, seed=11, dropout=0.0):
    schema = tr.session_schema(V, L)
    torch.manual_seed(0)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="mlm", embedding_dim_default=D)
    cfg = tr.XLNetConfig.build(D, 4, 2, total_seq_length=L, dropout=dropout, initializer_range=0.05)
    model = cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=True)).to(DEV).train()
    model.input_features.masking.seed = seed
    return schema, model


def test_recompute_head_falls_back_by_size_when_its_kernels_cannot_run(monkeypatch):
    """head_mode 'recompute' whose kernels cannot take the call (here: precision mode fp32) must not fall
    through to a materialised [N, V] tensor when the scores do not fit -- it takes the chunked head; where they fit it
    materialises.  Loss equal to the materialised mode either way."""
    import transformers4rec_amd as tr
    from transformers4rec_amd import ops
    from transformers4rec_amd.prediction_task import LazyPredictions

    B, L, V, D = 256, 20, 30000, 64
    ids = None

    def run(mode, limit_gb):
        nonlocal ids
        monkeypatch.setenv("T4R_HEAD_MODE", mode)
        monkeypatch.setenv("T4R_HEAD_AUTO_GB", limit_gb)
        schema, model = _model(tr, V, D, L)
        if ids is None:
            ids = tr.random_data_from_schema(schema, B, L, seed=3)["item_id"].to(DEV)
        with ops.precision("fp32"):
            out = model({"item_id": ids}, training=True)
            out["loss"].backward()
        torch.cuda.synchronize()
        return out, model.input_features.item_embedding_table.weight.grad.clone()

    ref, g_ref = run("materialize", "4")
    assert torch.is_tensor(ref["predictions"])
    small, g_small = run("recompute", "0.000001")          # the scores "do not fit": chunked head, nothing of size [N, V]
    assert isinstance(small["predictions"], LazyPredictions) and not small["predictions"].is_materialized
    fits, g_fits = run("recompute", "4")                   # they fit: materialised
    assert torch.is_tensor(fits["predictions"])
    for out, g in ((small, g_small), (fits, g_fits)):
        assert abs(float(out["loss"]) - float(ref["loss"])) < 2e-5
        torch.testing.assert_close(g, g_ref, rtol=1e-3, atol=1e-6 + 1e-4 * float(g_ref.abs().max()))


def test_auto_head_mode_respects_the_workspace_budget(monkeypatch):
    """`auto` takes the recomputing head only while its workspace fits T4R_HEAD_WS_GB"""
    import transformers4rec_amd as tr

    _, model = _model(tr, 30000, 64, 20)
    t = model.prediction_task
    t._training_call = True
    monkeypatch.setenv("T4R_HEAD_MODE", "auto")
    monkeypatch.setenv("T4R_HEAD_AUTO_GB", "0.000001")
    monkeypatch.setenv("T4R_HEAD_WS_GB", "16")
    assert t.resolve_head_mode(600, 30001) == "recompute"
    monkeypatch.setenv("T4R_HEAD_WS_GB", "0.000001")
    assert t.resolve_head_mode(600, 30001) == "fused"
    t._training_call = False                                  # evaluation calls: by size
    monkeypatch.setenv("T4R_HEAD_MODE", "recompute")
    assert t.resolve_head_mode(600, 30001) == "fused"
    monkeypatch.setenv("T4R_HEAD_AUTO_GB", "4")
    assert t.resolve_head_mode(600, 30001) == "materialize"


def _tools():
    import t4r_tools

    if not t4r_tools.available():
        pytest.skip("tools/bin/libt4r_tools.so not built (python -m transformers4rec_amd.build)")
    return t4r_tools


def test_tools_copy_kernel_copies():
    t = _tools()
    src = torch.randn(1 << 22, device=DEV)
    for mode in (0, 1):
        dst = torch.zeros_like(src)
        t.copy(dst, src, mode=mode)
        torch.cuda.synchronize()
        assert torch.equal(dst, src)


def test_occupier_holds_until_the_flag_and_never_beyond_its_bound():
    t = _tools()
    occ = t.Occupier(16, threads=256, lds_bytes=16 * 1024, max_us=2000)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    occ.start()
    occ.join()                       # nobody sets the flag: the time bound ends it
    e1.record()
    torch.cuda.synchronize()
    assert 1.5 < e0.elapsed_time(e1) < 50.0
    assert int(occ.seen.item()) == 16
    occ.start()
    occ.stop()                       # flag set at once: gone long before the bound
    occ.join()
    torch.cuda.synchronize()
    assert int(occ.seen.item()) == 32


@pytest.mark.parametrize("k,threads,lds", [(16, 256, 16 * 1024), (32, 512, 96 * 1024)])
def test_training_step_next_to_a_co_resident_collective_is_bit_identical(k, threads, lds):
    """one training step with k workgroups HOLDING CUs from the head's backward to the end of the backward
    pass (what an RCCL ring does while the table bucket is reduced under the body's backward): the token-tile kernels launch
    one workgroup per CU and must only get slower, never different -- every gradient bit-identical to the undisturbed step."""
    import transformers4rec_amd as tr

    t = _tools()
    B, L, V, D = 1024, 20, 20000, 128

    def run(occupied):
        schema, model = _model(tr, V, D, L, dropout=0.3)
        model.transformer_block.transformer.seed = 77
        ids = tr.random_data_from_schema(schema, B, L, seed=5)["item_id"].to(DEV)
        occ = t.Occupier(k if occupied else 0, threads=threads, lds_bytes=lds, max_us=20000)
        hook = tr.head_backward_hook(model, occ.start)
        out = model({"item_id": ids}, training=True)
        out["loss"].backward()
        occ.stop()
        occ.join()
        torch.cuda.synchronize()
        hook.remove()
        if occupied:
            assert int(occ.seen.item()) == k
        return float(out["loss"]), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    loss_a, g_a = run(False)
    loss_b, g_b = run(True)
    assert loss_a == loss_b and sorted(g_a) == sorted(g_b)
    for n in g_a:
        assert torch.equal(g_a[n], g_b[n]), n


# ------------------------------------------------------------------------------------------ the one-pass head forward
@pytest.mark.parametrize("N,V,D,smooth,T", [(700, 30001, 128, 0.0, 1.0), (333, 20017, 64, 0.1, 2.0), (2780, 100001, 128, 0.0, 1.0),
                                            (129, 9000, 32, 0.05, 1.0), (40, 70000, 96, 0.0, 0.5)])
def test_head_one_pass_forward_gives_logits_loss_and_dx(N, V, D, smooth, T):
    """t4r_head_split_logits_ce_dx (csrc/head_split.hip head_fwd_dx_kernel): logits, loss, lse and d X from
    ONE launch against (i) the two-pass form it replaces -- the logits bit for bit (same operand pieces, same product order),
    loss / lse / d X to rounding -- and (ii) fp64: transformers4rec/torch/model/prediction_task.py:648-671 + CrossEntropyLoss
    :446 and autograd's d X of them; d W from the SAME workspace (its per-item scales now come from the column maxima this
    forward leaves) against fp64 per item row."""
    from transformers4rec_amd import ops

    if not ops.head_split_fdx_supported(D):
        pytest.skip("one-pass head switched off (T4R_HEAD_FDX=0 / fp16 forms off)")
    g = torch.Generator().manual_seed(N + V)
    x = torch.randn(N, D, generator=g)
    # item norms over two decades (rare items: d W rows far below the tensor's maximum)
    W = torch.randn(V, D, generator=g) * (0.02 * torch.logspace(0, 2, V)[torch.randperm(V, generator=g)].unsqueeze(1) / 10)
    y = torch.randint(1, V, (N,), generator=g)
    xd, Wd, yd = x.to(DEV), W.to(DEV), y.to(DEV)
    ld = ops.pad_ld(V)
    # two-pass form
    ws0 = ops.head_split_prepare(xd, V)
    lg0, loss0, rows0, lse0 = ops.head_split_logits_ce(ws0, xd, Wd, yd, alpha=1.0 / T, label_smoothing=smooth, ldc=ld)
    one = torch.ones((), device=DEV)
    dx0 = ops.head_split_dx(ws0, lg0, lse0, yd, one, V, Wd, alpha=1.0 / T, label_smoothing=smooth)
    # one-pass form
    ws1 = ops.head_split_prepare(xd, V)
    lg1, loss1, rows1, lse1, dx1 = ops.head_split_logits_ce_dx(ws1, xd, Wd, yd, alpha=1.0 / T, label_smoothing=smooth, ldc=ld)
    dW1 = torch.zeros(V, D, device=DEV)
    ops.head_split_dw(ws1, lg1, lse1, yd, one, V, D, dW1, alpha=1.0 / T, label_smoothing=smooth, accumulate=True)
    torch.cuda.synchronize()
    assert torch.equal(lg0, lg1)
    assert float((lse0 - lse1).abs().max()) < 2e-6 * max(1.0, float(lse0.abs().max()))
    assert float((rows0 - rows1).abs().max()) < 4e-6 * max(1.0, float(rows0.abs().max()))
    assert abs(float(loss0) - float(loss1)) < 2e-6 * max(1.0, abs(float(loss0)))
    # fp64
    x64, W64 = x.double().requires_grad_(), W.double().requires_grad_()
    z = (x64 @ W64.t()) / T
    ref = torch.nn.functional.cross_entropy(z, y, label_smoothing=smooth)
    ref.backward()
    assert abs(float(loss1) - float(ref)) < 3e-6 * max(1.0, abs(float(ref)))
    assert float((lg1.double().cpu() - z.detach()).abs().max()) < 3e-6 * float(z.detach().abs().max())
    dmax = float(x64.grad.abs().max())
    assert float((dx1.double().cpu() - x64.grad).abs().max()) < 3e-6 * dmax
    assert float((dx0.double().cpu() - x64.grad).abs().max()) < 3e-6 * dmax
    # d W per item row (rows of rare items included): relative to the row's own largest entry
    gW = W64.grad
    rowmax = gW.abs().max(1).values
    live = rowmax > 0
    err = ((dW1.double().cpu() - gW).abs().max(1).values[live] / rowmax[live])
    assert float(err.max()) < 2e-5 and float(err.median()) < 3e-6
    if ops.head_split_dw_form(ws1) != 0:
        assert ops.head_split_dw_form(ws1) == 2          # the fp16 form with per-item scales, not the bf16-plane fallback
    # bit-reproducible
    ws2 = ops.head_split_prepare(xd, V)
    lg2, loss2, rows2, lse2, dx2 = ops.head_split_logits_ce_dx(ws2, xd, Wd, yd, alpha=1.0 / T, label_smoothing=smooth, ldc=ld)
    assert torch.equal(dx1, dx2) and torch.equal(lse1, lse2) and torch.equal(lg1, lg2)


def test_head_one_pass_forward_survives_extreme_scores():
    """rows whose largest score arrives late and far above the running reference (many bumps of the reference), all-equal
    scores, a label in the ragged last tile"""
    from transformers4rec_amd import ops

    N, V, D = 200, 12345, 128
    if not ops.head_split_fdx_supported(D):
        pytest.skip("one-pass head switched off")
    g = torch.Generator().manual_seed(3)
    W = 0.05 * torch.randn(V, D, generator=g)
    W[V // 2:] *= torch.linspace(1, 60, V - V // 2).unsqueeze(1)        # scores grow along the vocabulary
    x = torch.randn(N, D, generator=g)
    x[0] = 0.0                                                          # all-equal scores
    y = torch.randint(1, V, (N,), generator=g)
    y[1] = V - 1
    xd, Wd, yd = x.to(DEV), W.to(DEV), y.to(DEV)
    ws = ops.head_split_prepare(xd, V)
    lg, loss, rows, lse, dx = ops.head_split_logits_ce_dx(ws, xd, Wd, yd, ldc=ops.pad_ld(V))
    x64 = x.double().requires_grad_()
    z = x64 @ W.double().t()
    ref = torch.nn.functional.cross_entropy(z, y)
    ref.backward()
    assert bool(torch.isfinite(dx).all()) and bool(torch.isfinite(lse).all())
    assert abs(float(loss) - float(ref)) < 5e-6 * max(1.0, abs(float(ref)))
    # (peaked rows: d X = W[top] - W[y] to the rounding of table entries of magnitude ~3: an absolute 2e-7 |W| term)
    assert float((dx.double().cpu() - x64.grad).abs().max()) < 5e-6 * float(x64.grad.abs().max()) + 2e-7 * float(W.abs().max())


def test_cu_budget_changes_the_backward_tiles_not_the_gradients():
    """t4r_xlnet_set_cu_budget (what distributed.GradReducer sets while the table all-reduce holds CUs): the backward's
    token-tile kernels take 48-row tiles instead of 80-row ones at a budget of 240 CUs -- every per-row result (the gradient
    w.r.t. the layer input, hence the table gradient) stays bit-identical, the batch-reduced parameter gradients (LayerNorm,
    biases: per-workgroup partial sums, now grouped differently) agree to rounding."""
    import transformers4rec_amd as tr
    from transformers4rec_amd import ops

    B, L, V, D = 1024, 20, 20000, 128

    def run(budget):
        schema, model = _model(tr, V, D, L, dropout=0.3)
        model.transformer_block.transformer.seed = 77
        ids = tr.random_data_from_schema(schema, B, L, seed=5)["item_id"].to(DEV)
        out = model({"item_id": ids}, training=True)
        ops.xlnet_set_cu_budget(budget)
        try:
            out["loss"].backward()
            torch.cuda.synchronize()
        finally:
            ops.xlnet_set_cu_budget(0)
        return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    assert ops.xlnet_get_cu_budget() == 0
    g_a, g_b = run(0), run(240)
    assert sorted(g_a) == sorted(g_b)
    for n in g_a:
        scale = float(g_a[n].abs().max())
        assert float((g_a[n] - g_b[n]).abs().max()) <= 4e-6 * scale + 1e-12, n
    # the masked-item embedding and the table rows are sums over per-row results only
    tab = [n for n in g_a if n.endswith("item_embedding_table.weight") or n.endswith("item_id.weight")]
    assert tab


def test_soft_embedding_backward_is_bit_reproducible_and_accumulates():
    """round 5: the SoftEmbedding (+ LayerNorm) backward sums over the tokens without atomics (DPP wave sums, per-wave LDS
    slots, per-workgroup partial rows reduced in block order) -- at C3's size two calls give the same bits, a second call on
    the same buffers doubles them exactly, and the values match fp64 autograd of features/embedding.py:551-556 + LayerNorm."""
    from transformers4rec_amd import ops

    g = torch.Generator().manual_seed(0)
    T, K, D, W, col = 20480, 10, 8, 336, 320
    x = torch.rand(T, generator=g)
    pw, pb, tab = torch.randn(K, 1, generator=g), torch.randn(K, generator=g), torch.randn(K, D, generator=g)
    lw = 1 + 0.1 * torch.randn(D, generator=g)
    dy = torch.randn(T, W, generator=g)
    cu = lambda t: t.to(DEV).contiguous()
    args = (cu(dy), cu(x), cu(pw), cu(pb), cu(tab), cu(lw))

    def run(bufs=None):
        gr = bufs or [torch.zeros(t.shape, device=DEV) for t in (pw, pb, tab, lw, lw)]
        ops.soft_embedding_bwd(*args, gr[0], gr[1], gr[2], gr[3], gr[4], col=col)
        torch.cuda.synchronize()
        return gr

    a, b = run(), run()
    assert all(torch.equal(p, q) for p, q in zip(a, b))
    twice = run([t.clone() for t in a])
    assert all(torch.equal(t2, 2 * t1) for t1, t2 in zip(a, twice))
    ps = [t.double().requires_grad_() for t in (pw, pb, tab, lw)]
    wts = torch.softmax(x.double()[:, None] * ps[0][:, 0][None] + ps[1][None], -1)
    e = wts @ ps[2]
    out = torch.nn.functional.layer_norm(e, (D,), ps[3], torch.zeros(D, dtype=torch.float64), 1e-5)
    out.backward(dy[:, col:col + D].double())
    for got, p in zip(a[:4], ps):
        ref = p.grad
        assert float((got.double().cpu() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


# ------------------------------------------------------------------------------------------ one sort for several tables
@pytest.mark.parametrize("n,rows", [(20480, [100001, 1001, 501, 2001]), (777, [50, 3]), (4096, [1 << 20, 7, 7, 7, 300000])])
def test_one_sort_for_several_tables_equals_the_per_table_sorts(n, rows):
    """ops.sort_ids_multi (t4r_sort_ids_multi: the F tables of a multi-feature input block in one device sort) gives, per table,
    exactly the (keys, perm) of ops.sort_ids on that table alone -- padding ids, negative and out-of-range ids included."""
    from transformers4rec_amd import ops

    g = torch.Generator(device="cpu").manual_seed(n)
    ids = []
    for r in rows:
        x = torch.randint(-2, r + 3, (n,), generator=g)          # a few ids below 0 and beyond the table
        x[torch.rand(n, generator=g) < 0.2] = 0                   # padding
        ids.append(x.to(DEV))
    pads = [0] * len(rows)
    multi = ops.sort_ids_multi(ids, rows, pads)
    for f, r in enumerate(rows):
        k, p = ops.sort_ids(ids[f], r, 0)
        assert torch.equal(multi[f][0], k) and torch.equal(multi[f][1], p), f


def test_multi_feature_training_steps_are_bit_reproducible():
    """BASELINE configs[2] (item id + three categoricals + two SoftEmbeddings, concat, projection): three optimizer steps from
    the same seeds, twice -- identical losses and identical parameters bit for bit.  The projection's weight gradient was the
    last sum of that step taken with fp32 atomics in arrival order (ops.gemm_wgrad: split-K partials added in split order)."""
    sys.path.insert(0, ROOT)
    import bench

    dev = torch.device("cuda", 0)

    def run():
        tr, schema, model, dense, tables, opt = bench.build(dev, 0.3, config="c3")
        reducer, _ = bench.setup_data_parallel(tr, model, dense, tables, 1)
        model.input_features.masking.seed, model.transformer_block.transformer.seed = bench.rank_seeds(0)
        batches = [tr.random_data_from_schema(schema, bench.BATCH, bench.SEQ, seed=i, device=dev) for i in range(3)]
        model.train()
        step = bench.make_train_step(model, batches, reducer, opt)
        losses = [float(step(i)["loss"]) for i in range(3)]
        torch.cuda.synchronize()
        return losses, [f.data.clone() for f in opt.flats]

    la, pa = run()
    lb, pb = run()
    assert la == lb
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)


@pytest.mark.parametrize("D", [128, 64, 32])
def test_tiled_weight_planes_equal_the_element_wise_ones(D, monkeypatch):
    """layer_planes_tiled_kernel (32 x 32 tiles, transposed destinations through LDS) writes the same bytes as
    layer_planes_kernel: every plane of every weight matrix of a layer, both orientations, and the fp32 copies."""
    from transformers4rec_amd import ops

    torch.manual_seed(D)
    n, dh = 4, D // 4
    g32 = lambda *s, std=0.05: torch.randn(*s, device=DEV) * std
    prm = [g32(D, n, dh), g32(D, n, dh), g32(D, n, dh), g32(D, n, dh), g32(D, n, dh), g32(n, dh), g32(n, dh), 1 + g32(D), g32(D),
           g32(4 * D, D), g32(4 * D), g32(D, 4 * D), g32(D), 1 + g32(D), g32(D)]
    from transformers4rec_amd import _lib

    def planes_of(misaligned):
        # the tiled kernel takes a call whose sources are 16-byte aligned; one that is not falls to the element-wise kernel
        # (the product library has no switch for it: csrc/t4r_common.h t4r_exp_getenv)
        src = prm
        if misaligned:
            src = []
            for t in prm:
                buf = torch.zeros(t.numel() + 8, device=DEV)
                off = 2 + (4 - (buf.data_ptr() // 4) % 4) % 4          # data_ptr % 16 == 8
                v = buf[off: off + t.numel()].view(t.shape)
                v.copy_(t)
                assert v.data_ptr() % 16 == 8
                src.append(v)
        planes = torch.zeros(_lib.load().t4r_xlnet_layer_planes_floats(D), device=DEV, dtype=torch.float32)   # unwritten gaps stay 0
        ptrs, _keep = _lib.ptr_array([t.data_ptr() for t in src])
        _lib.call("t4r_xlnet_layer_prepare", torch.cuda.current_stream().cuda_stream, ptrs, D, planes.data_ptr())
        torch.cuda.synchronize()
        return planes

    a, b = planes_of(True), planes_of(False)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert int((a.view(torch.int32) != 0).sum()) > a.numel() // 4        # the comparison is not of two empty buffers
