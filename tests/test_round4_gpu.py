"""GPU tests of round 4.

  * VERDICT r3 weak #1 / next #1a: the head's weight gradient d W (csrc/head_split.hip; reference chain
    transformers4rec/torch/model/prediction_task.py:648-671 + CrossEntropyLoss :446 through autograd) checked PER ROW
    (one row = one item) against fp64 with item norms spread over two decades -- the assertion the norm-wise tests are
    blind to -- in both kernel forms, with the form that ran asserted through the caller-owned note
    (include/t4r_hip.h: t4r_head_note, t4r_head_note_dw_form);
  * the same row-wise check on the item-table gradient of a whole BASELINE configs[1] step, after a few Adam steps and
    with item norms spread over two decades.
"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

BUCKETS = ((1e-1, 1e1), (1e-2, 1e-1), (1e-3, 1e-2), (1e-4, 1e-3), (1e-5, 1e-4), (0.0, 1e-5))


def _row_errors(dW, ref):
    """relative error of every row against ITS OWN largest entry, and the rows' magnitude relative to the largest row"""
    rmax = ref.abs().amax(dim=1)
    err = (dW.double() - ref).abs().amax(dim=1) / rmax.clamp_min(1e-300)
    return err, rmax / rmax.max()


def _assert_rowwise(err, rel, tol, what):
    seen = 0
    for lo, hi in BUCKETS:
        m = (rel >= lo) & (rel < hi)
        n = int(m.sum())
        if n:
            seen += 1
            worst = float(err[m].max())
            assert worst <= tol, f"{what}: rows with max |row| in [{lo:.0e}, {hi:.0e}) of the largest ({n} rows): " \
                                 f"relative row error {worst:.2e} > {tol:.0e}"
    return seen


@pytest.mark.parametrize("form", [2, 1])
def test_head_dw_rows_match_fp64_in_every_magnitude_bucket(form):
    """N = 2 780 label rows, V = 100 001 items, D = 128 (the shape of BASELINE configs[1]); item norms over two decades,
    so most items are rare and their gradient rows lie 1e-5 and more below the tensor's largest entry.  form 2: two-way
    fp16 split with per-item scales (the default when the forward left its column maxima); form 1: three bf16 planes
    (what runs without a note).  Tolerance: 5e-6 of each ROW's own largest entry."""
    from transformers4rec_amd import ops

    N, V, D = 2780, 100001, 128
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(N, D, device=DEV, generator=g)
    W = torch.randn(V, D, device=DEV, generator=g) * (0.05 + 0.45 * torch.rand(V, 1, device=DEV, generator=g) ** 3)
    labels = torch.randint(0, V, (N,), device=DEV, generator=g)
    gout = torch.tensor(1.0, device=DEV)
    ws = ops.head_split_prepare(x, V)
    logits, _, _, lse = ops.head_split_logits_ce(ws, x, W, labels, ldc=ops.pad_ld(V))
    p = torch.softmax(logits.double(), dim=1)
    p[torch.arange(N, device=DEV), labels] -= 1.0
    ref = (p / N).t() @ x.double()
    del p
    if form == 1:      # a fresh (zeroed) note: the backward product knows nothing about the forward
        ws.t4r_note = torch.zeros(8, dtype=torch.int64)
    assert ops.head_split_dw_form(ws) == 0
    dW = torch.full((V, D), float("nan"), device=DEV)
    ops.head_split_dw(ws, logits, lse, labels, gout, V, D, dW, accumulate=False)
    assert ops.head_split_dw_form(ws) == form, "the d W kernel form that ran is not the one under test"
    err, rel = _row_errors(dW, ref)
    assert int((rel < 1e-4).sum()) > V // 4, "the setup must contain many rare-item rows"
    assert _assert_rowwise(err, rel, 5e-6, f"head d W form {form}") >= 4
    # accumulate = True adds to what is there, in the same form
    dW2 = dW.clone()
    ops.head_split_dw(ws, logits, lse, labels, gout, V, D, dW2, accumulate=True)
    assert ops.head_split_dw_form(ws) == form
    err2, _ = _row_errors(dW2 * 0.5, ref)
    _assert_rowwise(err2, rel, 5e-6, f"head d W form {form} (accumulated)")


def test_head_note_is_per_forward_not_process_wide():
    """more forwards in flight than any ring could hold (gradient accumulation over micro-batches, the chunked head):
    every backward still finds ITS forward's note -- it travels with the workspace -- and runs the per-item-scale form"""
    from transformers4rec_amd import ops

    V, D = 20001, 64
    g = torch.Generator(device=DEV).manual_seed(3)
    W = torch.randn(V, D, device=DEV, generator=g) * 0.2
    saved = []
    for i in range(40):
        N = 700 + 8 * i
        x = torch.randn(N, D, device=DEV, generator=g)
        labels = torch.randint(0, V, (N,), device=DEV, generator=g)
        ws = ops.head_split_prepare(x, V)
        logits, _, _, lse = ops.head_split_logits_ce(ws, x, W, labels, ldc=ops.pad_ld(V))
        saved.append((x, labels, ws, logits, lse))
    gout = torch.tensor(1.0, device=DEV)
    for x, labels, ws, logits, lse in saved:
        dW = torch.empty(V, D, device=DEV)
        ops.head_split_dw(ws, logits, lse, labels, gout, V, D, dW, accumulate=False)
        assert ops.head_split_dw_form(ws) == 2
    x, labels, ws, logits, lse = saved[0]
    N = x.shape[0]
    p = torch.softmax(logits.double(), dim=1)
    p[torch.arange(N, device=DEV), labels] -= 1.0
    ref = (p / N).t() @ x.double()
    dW = torch.empty(V, D, device=DEV)
    ops.head_split_dw(ws, logits, lse, labels, gout, V, D, dW, accumulate=False)
    err, rel = _row_errors(dW, ref)
    _assert_rowwise(err, rel, 5e-6, "first of 40 forwards in flight")


def test_full_size_table_gradient_rows_after_adam_steps():
    """BASELINE configs[1] at full size (100 001 x 128 table, 4-layer XLNet, batch 1024 x 20, MLM, tied head, dropout 0):
    the item table is rescaled so that item norms cover a decade (logits of +-30: beyond that fp32 softmax itself is only
    good to |z| * 6e-8 * ln 2 per entry, see the next test), three Adam steps are taken, then ONE step's table gradient is
    checked per row.  For items that do not occur in the batch the table gradient IS the head's d W row
    (the lookup scatter does not touch them): fp64 reference from the step's own logits and head input rows."""
    import transformers4rec_amd as tr

    torch.manual_seed(0)
    B, L, V, D = 1024, 20, 100_000, 128
    schema = tr.session_schema(V, L)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="mlm", embedding_dim_default=D)
    cfg = tr.XLNetConfig.build(D, 4, 4, total_seq_length=L, dropout=0.0)
    model = cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=True))
    model.to(DEV)
    table = model.input_features.item_embedding_table.weight
    with torch.no_grad():
        g = torch.Generator(device=DEV).manual_seed(5)
        table.mul_(1.0 + 9.0 * torch.rand(V + 1, 1, device=DEV, generator=g) ** 3)
    dense, tables = tr.flatten_model(model)
    opt = tr.FusedAdam([dense, tables], lr=1e-2)
    table = model.input_features.item_embedding_table.weight
    hidden = {}
    hook = model.transformer_block.register_forward_hook(lambda m, a, out: hidden.__setitem__("h", out.detach()))
    for step in range(4):
        data = tr.random_data_from_schema(schema, B, L, seed=20 + step)
        ids = data["item_id"].to(DEV)
        out = model({"item_id": ids}, training=True)
        out["loss"].backward()
        # (with a x100 spread the logits reach +-250 and an item far below every row's lse has a d W bound that
        # underflows: its power-of-two scale used to overflow to inf -- NaN in the whole table gradient, found here)
        assert bool(torch.isfinite(table.grad).all()), f"non-finite table gradient at step {step}"
        if step < 3:
            opt.step()
    hook.remove()
    norms = table.detach().norm(dim=1)
    assert float(norms.max() / norms.min()) > 5.0
    mask = model.input_features.masking.mask_schema
    X = hidden["h"][mask].double()                                  # the head's input rows, row-major (b, l) order
    labels = out["labels"]
    N = X.shape[0]
    assert N == labels.shape[0] and N > 2000
    p = torch.softmax(out["predictions"].detach().double(), dim=1)
    p[torch.arange(N, device=DEV), labels] -= 1.0
    ref = (p / N).t() @ X
    del p
    absent = torch.ones(V + 1, dtype=torch.bool, device=DEV)
    absent[ids.flatten()] = False
    assert int(absent.sum()) > V // 2
    err, rel = _row_errors(table.grad[absent], ref[absent])
    held = rel >= 1e-25              # rows fp32 can hold next to the largest (p ~ 1e-30 and below is zero in fp32)
    assert int(held.sum()) > V // 4
    assert _assert_rowwise(err[held], rel[held], 2e-5, "item-table gradient rows of absent items") >= 3


def test_head_dw_with_logits_of_hundreds_stays_finite_and_right():
    """logits of +-300 (an untrained model with large item norms): an item whose best logit lies hundreds below every
    row's lse has a gradient bound that underflows; its power-of-two position must saturate, not overflow.  Rows whose
    fp64 gradient is representable are still right per row; the others are (correctly) zero."""
    from transformers4rec_amd import ops

    N, V, D = 1500, 30001, 128
    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn(N, D, device=DEV, generator=g)
    W = torch.randn(V, D, device=DEV, generator=g) * (0.2 + 6.0 * torch.rand(V, 1, device=DEV, generator=g) ** 2)
    labels = torch.randint(0, V, (N,), device=DEV, generator=g)
    gout = torch.tensor(1.0, device=DEV)
    ws = ops.head_split_prepare(x, V)
    logits, loss, _, lse = ops.head_split_logits_ce(ws, x, W, labels, ldc=ops.pad_ld(V))
    assert float(logits.abs().max()) > 150.0 and bool(torch.isfinite(loss))
    p = torch.softmax(logits.double(), dim=1)
    p[torch.arange(N, device=DEV), labels] -= 1.0
    ref = (p / N).t() @ x.double()
    del p
    dW = torch.full((V, D), float("nan"), device=DEV)
    ops.head_split_dw(ws, logits, lse, labels, gout, V, D, dW, accumulate=False)
    assert ops.head_split_dw_form(ws) == 2
    assert bool(torch.isfinite(dW).all())
    dX = ops.head_split_dx(ws, logits, lse, labels, gout, V, W)
    assert bool(torch.isfinite(dX).all())
    gmax = float(ref.abs().max())
    # at |z| = 300 an fp32 exp argument carries 300 * log2(e) * 6e-8 = 2.6e-5 of absolute error: ANY fp32 softmax
    # gradient (the reference's included) is good to ~2e-5 relative here, whatever the products do
    assert float((dW.double() - ref).abs().max()) < 1e-4 * gmax
    # per row: relative to the row where the row matters, absolute below that.  An item's scale comes from a BOUND on its
    # column (best logit against the smallest lse of any row); with lse spread over hundreds between rows the bound can
    # sit far above the entries, whose fp16 pieces then run out of range: the row keeps an absolute accuracy of ~1e-11 of
    # the largest gradient -- far below what Adam's eps (1e-8) lets through -- instead of a relative one
    rmax = ref.abs().amax(dim=1)
    aerr = (dW.double() - ref).abs().amax(dim=1)
    assert bool((aerr <= 2e-4 * rmax + 1e-9 * gmax).all()), float((aerr - 2e-4 * rmax).max() / gmax)


# ------------------------------------------------------------------------------------------ the recomputing head
@pytest.mark.parametrize("N,V,D,alpha,smooth", [(77, 1000, 64, 1.0, 0.0), (130, 999, 96, 2.0, 0.0), (200, 5000, 128, 0.5, 0.1),
                                                 (33, 257, 32, 1.0, 0.0), (2780, 100001, 128, 1.0, 0.0)])
def test_recomputing_head_matches_fp64_and_the_materialised_kernels(N, V, D, alpha, smooth):
    """csrc/head_split.hip, round 4: cross-entropy without a logits tensor + the two backward products on recomputed score
    tiles, against fp64 (loss, lse, d X, d W -- d W also per row) and against the materialised kernels they replace
    (reference chain: prediction_task.py:648-671 + CrossEntropyLoss :446 and their autograd)"""
    from transformers4rec_amd import ops

    if not ops.head_split_recompute_supported(D):
        pytest.skip("two-way fp16 form switched off")
    g = torch.Generator(device=DEV).manual_seed(N + V)
    x = torch.randn(N, D, device=DEV, generator=g)
    W = torch.randn(V, D, device=DEV, generator=g) * (0.05 + 0.45 * torch.rand(V, 1, device=DEV, generator=g) ** 3)
    labels = torch.randint(0, V, (N,), device=DEV, generator=g)
    gout = torch.tensor(1.7, device=DEV)
    ws = ops.head_split_prepare(x, V)
    loss, rows, lse = ops.head_split_ce(ws, x, W, labels, alpha=alpha, label_smoothing=smooth)
    lg64 = alpha * (x.double() @ W.double().t())
    ref_rows = torch.nn.functional.cross_entropy(lg64, labels, reduction="none", label_smoothing=smooth)
    assert float((rows.double() - ref_rows).abs().max()) < 2e-5
    assert float((lse.double() - torch.logsumexp(lg64, 1)).abs().max()) < 2e-5
    assert abs(float(loss) - float(ref_rows.mean())) < 2e-5
    p = torch.softmax(lg64, dim=1)
    onehot = torch.zeros_like(p)
    onehot[torch.arange(N, device=DEV), labels] = 1.0
    G = (1.7 / N) * (p - (1 - smooth) * onehot - smooth / V)
    dX64, dW64 = alpha * (G @ W.double()), alpha * (G.t() @ x.double())
    del p, onehot, G
    dX = ops.head_split_dx_rc(ws, x, W, lse, labels, gout, alpha=alpha, label_smoothing=smooth)
    dW0 = torch.randn(V, D, device=DEV, generator=g) * float(dW64.abs().max())
    dW = dW0.clone()
    ops.head_split_dw_rc(ws, W, lse, labels, gout, dW, alpha=alpha, label_smoothing=smooth, accumulate=True)
    assert ops.head_split_dw_form(ws) == 3
    dWn = torch.full((V, D), float("nan"), device=DEV)
    ops.head_split_dw_rc(ws, W, lse, labels, gout, dWn, alpha=alpha, label_smoothing=smooth, accumulate=False)
    assert float((dX.double() - dX64).abs().max()) < 5e-6 * float(dX64.abs().max())
    assert float((dWn.double() - dW64).abs().max()) < 5e-6 * float(dW64.abs().max())
    assert float((dW.double() - dW0.double() - dW64).abs().max()) < 1e-5 * float(dW64.abs().max())
    if smooth == 0.0:           # per row: every item's gradient row to 1e-5 of ITS largest entry (smoothing adds a floor of eps / V to every row)
        err, rel = _row_errors(dWn, dW64)
        _assert_rowwise(err, rel, 1e-5, "recomputed d W")
    # bit-reproducible, and the same numbers as the materialised kernels to fp32 rounding
    dX2 = ops.head_split_dx_rc(ws, x, W, lse, labels, gout, alpha=alpha, label_smoothing=smooth)
    dW2 = torch.empty_like(dWn)
    ops.head_split_dw_rc(ws, W, lse, labels, gout, dW2, alpha=alpha, label_smoothing=smooth, accumulate=False)
    assert torch.equal(dX, dX2) and torch.equal(dWn, dW2)
    ws_m = ops.head_split_prepare(x, V)
    logits, loss_m, rows_m, lse_m = ops.head_split_logits_ce(ws_m, x, W, labels, alpha=alpha, label_smoothing=smooth, ldc=ops.pad_ld(V))
    assert torch.equal(lse, lse_m) and float((rows - rows_m).abs().max()) < 1e-6 * max(1.0, float(rows_m.abs().max()))
    dX_m = ops.head_split_dx(ws_m, logits, lse_m, labels, gout, V, W, alpha=alpha, label_smoothing=smooth)
    assert float((dX - dX_m).abs().max()) < 2e-6 * float(dX64.abs().max())


def test_training_step_takes_the_recomputing_head_and_keeps_its_numbers(monkeypatch):
    """the module mirror's training step at a head_split.hip shape with head_mode "recompute" (no [N, V] tensor; `predictions`
    is lazy; `auto` takes it when the scores would not fit), evaluation keeps the materialised scores; loss and every gradient
    equal the materialised mode"""
    import transformers4rec_amd as tr
    from transformers4rec_amd import ops

    B, L, V, D = 256, 20, 30000, 64
    schema = tr.session_schema(V, L)

    def run(mode):
        monkeypatch.setenv("T4R_HEAD_MODE", mode)
        torch.manual_seed(0)
        inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="mlm", embedding_dim_default=D)
        cfg = tr.XLNetConfig.build(D, 4, 2, total_seq_length=L, dropout=0.0, initializer_range=0.05)
        model = cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=True)).to(DEV).train()
        model.input_features.masking.seed = 11
        ids = tr.random_data_from_schema(schema, B, L, seed=3)["item_id"].to(DEV)
        out = model({"item_id": ids}, training=True)
        out["loss"].backward()
        torch.cuda.synchronize()
        return model, out, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    m_a, out_a, g_a = run("recompute")
    assert m_a.prediction_task.resolve_head_mode(out_a["labels"].numel(), V + 1) == "recompute"
    from transformers4rec_amd.prediction_task import LazyPredictions
    assert isinstance(out_a["predictions"], LazyPredictions) and not out_a["predictions"].is_materialized
    m_m, out_m, g_m = run("materialize")
    assert torch.equal(out_a["labels"], out_m["labels"])
    assert abs(float(out_a["loss"]) - float(out_m["loss"])) < 2e-6
    torch.testing.assert_close(out_a["predictions"].materialize(), out_m["predictions"], rtol=1e-5, atol=1e-5)
    assert sorted(g_a) == sorted(g_m)
    for k in g_a:
        torch.testing.assert_close(g_a[k], g_m[k], rtol=1e-4, atol=1e-7 + 2e-6 * float(g_m[k].abs().max()), msg=lambda m, k=k: f"{k}: {m}")
    # auto: recompute exactly when the scores would not fit the materialisation limit
    monkeypatch.setenv("T4R_HEAD_MODE", "auto")
    monkeypatch.setenv("T4R_HEAD_AUTO_GB", "0.001")
    assert m_a.prediction_task.resolve_head_mode(out_a["labels"].numel(), V + 1) == "recompute"
    monkeypatch.setenv("T4R_HEAD_AUTO_GB", "4")
    assert m_a.prediction_task.resolve_head_mode(out_a["labels"].numel(), V + 1) == "materialize"
    monkeypatch.setenv("T4R_HEAD_MODE", "recompute")
    m_a.eval()
    with torch.no_grad():
        ev = m_a({"item_id": tr.random_data_from_schema(schema, B, L, seed=4)["item_id"].to(DEV)}, testing=True)
    assert torch.is_tensor(ev["predictions"])            # evaluation: materialised scores
