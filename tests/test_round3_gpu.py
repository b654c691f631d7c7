"""GPU parity of the rows closed in round 3 (VERDICT r2 "missing" 1, 2, 4, 5):
  * C1 = BASELINE configs[0]: the first 100 sessions of the reference's own testing data, fed as a PARQUET file +
    schema.json through ParquetSessionLoader / Schema.from_json -> TabularSequenceFeatures -> XLNet (d 64, 2 layers,
    4 heads) -> tied NextItemPredictionTask, against the fixture the unmodified reference produced from the same rows
    (train: labels bit-exact, loss / scores / every gradient; eval; inference) -- and the reference's training loop
    contract (Model.fit over the loader, torch/model/base.py:669-739);
  * a3: EmbeddingFeatures' EmbeddingBag branch (mean | sum | sqrtn; [B], [B, K], (values, offsets)) against the reference
    fixtures and the oracle;
  * the default metric set NDCG / AvgPrecision / Recall @k and the `metrics=` argument against the reference's values.
"""
import json
import os

import numpy as np
import pytest
import torch

import golden_utils as gu
import t4r_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = dict(rtol=1e-4, atol=5e-5)


def close(a, b, **kw):
    t = dict(TOL)
    t.update(kw)
    torch.testing.assert_close(a.detach().cpu(), b.detach().cpu(), **t)


# ------------------------------------------------------------------------------------------ C1
ITEM = "item_id/list"


def _c1_files(d, tmp_path):
    """the fixture's wire-form rows -> data.parquet + schema.json, the two files configs[0] names"""
    import pyarrow as pa
    import pyarrow.parquet as pq

    vals, offs = d["in_ragged/" + ITEM + "__values"], d["in_ragged/" + ITEM + "__offsets"]
    col = pa.ListArray.from_arrays(pa.array(offs, type=pa.int32()), pa.array(vals, type=pa.int64()))
    sess = pa.array(np.arange(len(offs) - 1, dtype=np.int64))
    pq.write_table(pa.table({"session_id": sess, ITEM: col}), str(tmp_path / "data.parquet"))
    (tmp_path / "schema.json").write_text(str(d["meta/schema_json"]))
    return str(tmp_path / "data.parquet"), str(tmp_path / "schema.json")


def _c1_model(d, schema):
    import transformers4rec_amd as tr

    L = int(d["meta/L"])
    inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="mlm",
                                                    embedding_dim_default=int(d["meta/d_model"]))
    cfg = tr.XLNetConfig.build(int(d["meta/d_model"]), int(d["meta/n_head"]), int(d["meta/n_layer"]),
                               total_seq_length=L, dropout=0.0)
    model = cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=True))
    sd = gu.section(d, "p/")
    own = model.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            assert k in own and own[k].shape == v.shape, k
            own[k].copy_(v)
    return model.to(DEV)


def _c1_scores(d, scores):
    s = scores.detach().float().cpu()
    cols = gu.t(d["sel/cols"])
    close(s[:, cols], gu.t(d["out/predictions_sel"]))
    assert float((s[:, cols] - gu.t(d["out/predictions_sel"])).abs().max()) < 1e-3          # north_star gate
    torch.testing.assert_close(torch.logsumexp(s.double(), -1), gu.t(d["out/predictions_lse"]), rtol=1e-6, atol=2e-5)
    torch.testing.assert_close(s.double().sum(-1), gu.t(d["out/predictions_rowsum"]), rtol=1e-5, atol=5e-2)
    assert torch.equal(s.argmax(-1), gu.t(d["out/predictions_argmax"]))


def test_c1_reference_testing_data_through_parquet_loader(tmp_path):
    import transformers4rec_amd as tr

    d = gu.c1_load()
    pq_path, js_path = _c1_files(d, tmp_path)
    schema = tr.Schema.from_json(js_path).select_by_name([ITEM])
    assert schema.item_id_column_name == ITEM
    model = _c1_model(d, schema)
    V = int(d["meta/V"])
    assert model.input_features.item_embedding_table.weight.shape[0] == V == 51997
    loader = tr.ParquetSessionLoader(pq_path, batch_size=int(d["meta/rows"]), max_sequence_length=int(d["meta/L"]),
                                     schema=schema, device=DEV)
    (batch,) = list(loader)
    assert torch.equal(batch[ITEM].cpu(), gu.t(d["in/" + ITEM]))          # == the reference's pad_batch of the same rows
    masking = model.input_features.masking
    masking.set_draws(gu.t(d["draw/bern"]).to(DEV).to(torch.uint8), gu.t(d["draw/j1"]).to(DEV), gu.t(d["draw/j2"]).to(DEV))
    cap = {}
    model.input_features.register_forward_hook(lambda m, i, o: cap.__setitem__("emb", o.detach().clone()))
    model.transformer_block.register_forward_hook(lambda m, i, o: cap.__setitem__("hid", o.detach().clone()))
    out = model(batch, training=True)
    assert torch.equal(masking.mask_schema.cpu(), gu.t(d["out/mask_schema"]))
    assert torch.equal(masking.masked_targets.cpu(), gu.t(d["out/masked_targets"]))
    assert torch.equal(out["labels"].cpu(), gu.t(d["out/labels"]))
    close(cap["emb"], gu.t(d["out/inputs_embeds"]))
    close(cap["hid"], gu.t(d["out/hidden"]))
    _c1_scores(d, out["predictions"])
    close(out["loss"], gu.t(d["out/loss"]))
    assert abs(float(out["loss"]) - float(d["out/loss"])) < 1e-3
    out["loss"].backward()
    g = gu.section(d, "g/")
    named = dict(model.named_parameters())
    for k, ref in g.items():
        close(named[k].grad, ref, rtol=2e-4, atol=1e-4, msg=lambda m, k=k: f"{k}: {m}")
    key = str(d["meta/table_key"])
    gt = named[key].grad.cpu()
    close(gt[gu.t(d["sel/cols"])], gu.t(d["gsel/" + key]), rtol=2e-4, atol=1e-4)
    torch.testing.assert_close(gt.double().sum(0), gu.t(d["gsum/" + key]), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(gt.double().abs().sum(0), gu.t(d["gabs/" + key]), rtol=1e-4, atol=1e-4)
    # evaluation (last item) and inference on the same rows
    model.eval()
    e, f = gu.c1_load("c1_yoochoose_eval"), gu.c1_load("c1_yoochoose_infer")
    with torch.no_grad():
        oe = model(batch, testing=True)
        assert torch.equal(oe["labels"].cpu(), gu.t(e["out/labels"]))
        _c1_scores(e, oe["predictions"])
        close(oe["loss"], gu.t(e["out/loss"]))
        _c1_scores(f, model(batch))


def test_c1_fit_loop_over_the_loader(tmp_path):
    """the plain training loop the reference runs for configs[0] (Model.fit: for batch in dataloader: forward,
    loss.backward, optimizer.step -- torch/model/base.py:669-739): one epoch over the 100 sessions in 4 batches with
    torch.optim.Adam on model.parameters(); the loss falls and stays finite"""
    import transformers4rec_amd as tr

    d = gu.c1_load()
    pq_path, js_path = _c1_files(d, tmp_path)
    schema = tr.Schema.from_json(js_path).select_by_name([ITEM])
    model = _c1_model(d, schema)
    model.train()
    loader = tr.ParquetSessionLoader(pq_path, batch_size=25, max_sequence_length=20, schema=schema, device=DEV, shuffle=True)
    assert len(loader) == 4 and len(loader.dataset) == 100 and loader._batch_size == 25
    opt = torch.optim.Adam(model.parameters(), lr=5e-3)
    losses = []
    for epoch in range(3):
        for batch in loader:
            opt.zero_grad(set_to_none=False)
            out = model(batch, training=True)
            out["loss"].backward()
            opt.step()
            losses.append(float(out["loss"]))
    assert all(np.isfinite(losses)) and len(losses) == 12
    assert np.mean(losses[-4:]) < np.mean(losses[:4])


# ------------------------------------------------------------------------------------------ a3: EmbeddingBag
def _bag_module(d, comb, aggregation=None):
    import transformers4rec_amd as tr

    cfg = {k: tr.FeatureConfig(tr.TableConfig(int(d["p/" + k].shape[0]), int(d["p/" + k].shape[1]), name=k, combiner=comb))
           for k in ("genres", "tags", "country")}
    mod = tr.EmbeddingFeatures(cfg, aggregation=aggregation)
    with torch.no_grad():
        for k in cfg:
            mod.embedding_tables[k].weight.copy_(gu.t(d["p/" + k]))
    return mod.to(DEV)


def _bag_inputs(d):
    return {"genres": gu.t(d["in/genres"]).to(DEV), "country": gu.t(d["in/country"]).to(DEV),
            "tags": (gu.t(d["in/tags_values"]).to(DEV).unsqueeze(-1), gu.t(d["in/tags_offsets"]).to(DEV).unsqueeze(-1))}


@pytest.mark.parametrize("comb", ["mean", "sum"])
def test_embedding_bag_matches_reference(comb):
    d = gu.load(f"embedding_bag_{comb}")
    mod = _bag_module(d, comb)
    assert sorted(mod.state_dict()) == [f"embedding_tables.{k}.weight" for k in ("country", "genres", "tags")]
    out = mod(_bag_inputs(d))
    for k in ("genres", "tags", "country"):
        close(out[k], gu.t(d["out/" + k]), rtol=1e-5, atol=1e-6)
    assert float(out["tags"][0].abs().sum()) == 0.0                      # empty bag -> zero row
    sum((out[k] * gu.t(d["c/" + k]).to(DEV)).sum() for k in out).backward()
    for k in out:
        close(mod.embedding_tables[k].weight.grad, gu.t(d["g/" + k]), rtol=1e-5, atol=1e-6)
    mod.check_ids()
    # concat aggregation = the same rows side by side in sorted-name order, written by the gather itself
    modc = _bag_module(d, comb, aggregation="concat")
    oc = modc(_bag_inputs(d))
    ref = torch.cat([gu.t(d["out/" + k]) for k in ("country", "genres", "tags")], -1)
    close(oc, ref, rtol=1e-5, atol=1e-6)
    (oc * torch.cat([gu.t(d["c/" + k]) for k in ("country", "genres", "tags")], -1).to(DEV)).sum().backward()
    for k in out:
        close(modc.embedding_tables[k].weight.grad, gu.t(d["g/" + k]), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("comb", ["mean", "sum", "sqrtn"])
@pytest.mark.parametrize("dim", [6, 64, 132, 320])
def test_embedding_bag_vs_oracle_random(comb, dim):
    """ragged + matrix bags on seeded inputs (long bags, empty bags, odd / wide row widths), forward and table gradient;
    twice: the gradient is bit-reproducible (sorted scatter, no atomics)"""
    from transformers4rec_amd import ops

    g = torch.Generator().manual_seed(dim)
    V, B = 1000, 257
    table = (0.1 * torch.randn(V, dim, generator=g))
    lens = torch.randint(0, 40, (B,), generator=g)
    lens[3] = 0
    lens[B - 1] = 0                                                        # empty LAST bag
    offs = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)])[:-1]
    vals = torch.randint(0, V, (int(lens.sum()),), generator=g)
    mat = torch.randint(0, V, (B, 7), generator=g)
    c = torch.randn(B, dim, generator=g)
    for kw_o, kw_h in ((dict(values=vals, offsets=offs), (vals.to(DEV), offs.to(DEV))), (dict(ids=mat), (mat.to(DEV), None))):
        t_o = table.clone().requires_grad_()
        ref = O.embedding_bag(t_o, combiner=comb, **kw_o)
        (ref * c).sum().backward()
        td = table.to(DEV)
        out = ops.embedding_bag_fwd(td, kw_h[0], kw_h[1], comb)
        close(out, ref, rtol=1e-5, atol=1e-6)
        grads = []
        for _ in range(2):
            gt = torch.zeros_like(td)
            n = kw_h[0].numel()
            rows = ops.embedding_bag_bwd_rows(c.to(DEV), n, dim, kw_h[1], 0 if kw_h[1] is not None else 7, comb)
            ops.scatter_rows_sorted(gt, kw_h[0].reshape(-1), rows, padding_idx=-1)
            grads.append(gt)
        close(grads[0], t_o.grad, rtol=1e-5, atol=2e-6)
        assert torch.equal(grads[0], grads[1])


def test_embedding_bag_flags_out_of_range_ids():
    import transformers4rec_amd as tr

    mod = tr.EmbeddingFeatures({"a": tr.FeatureConfig(tr.TableConfig(10, 8, combiner="sum", name="a"))}).to(DEV)
    mod({"a": torch.tensor([[1, 2], [3, 10]], device=DEV)})
    with pytest.raises(IndexError):
        mod.check_ids()
    with pytest.raises(Exception):
        mod({"a": torch.tensor([[1, 2]])})              # CPU tensor: there is no CPU path


# ------------------------------------------------------------------------------------------ ranking metrics
def test_default_metrics_match_reference_values():
    import transformers4rec_amd as tr

    d = gu.load("ranking_metrics")
    ks = [int(k) for k in d["meta/top_ks"]]
    scores, labels = gu.t(d["in/scores"]).to(DEV), gu.t(d["in/labels"]).to(DEV)
    task = tr.NextItemPredictionTask(weight_tying=True, top_ks=ks)
    assert [m.name for m in task.metrics] == ["ndcg_at", "avg_precision_at", "recall_at"]   # DEFAULT_METRICS order
    rows = task.calculate_metrics(scores[:40].contiguous(), labels[:40])
    for name in ("ndcg_at", "avg_precision_at", "recall_at"):
        for j, k in enumerate(ks):
            close(rows[f"{name}_{k}"], gu.t(d[f"out/rows/{name}"])[:40, j], rtol=1e-6, atol=1e-6)
    task.calculate_metrics(scores[40:].contiguous(), labels[40:])
    agg = task.compute_metrics()
    for name in ("ndcg_at", "avg_precision_at", "recall_at"):
        for j, k in enumerate(ks):
            assert abs(agg[f"next-item/{name}_{k}"] - float(d[f"out/mean/{name}"][j])) < 1e-6, (name, k)
    # the metrics= argument: registry names and descriptor objects
    task2 = tr.NextItemPredictionTask(weight_tying=True, metrics=[tr.PrecisionAt(top_ks=ks), "dcg_at"], top_ks=ks)
    rows = task2.calculate_metrics(scores, labels)
    for name in ("precision_at", "dcg_at"):
        for j, k in enumerate(ks):
            close(rows[f"{name}_{k}"], gu.t(d[f"out/rows/{name}"])[:, j], rtol=1e-6, atol=1e-6)
    with pytest.raises(NotImplementedError):
        tr.NextItemPredictionTask(metrics=["auc"])
    task2.reset_metrics()
    assert task2.compute_metrics() == {}
