"""Pins the oracle (oracle/t4r_oracle.py) against fixtures produced by the unmodified
reference (oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import golden_utils as gu
import t4r_oracle as O

TOL = dict(rtol=2e-5, atol=2e-5)


def _inputs(d):
    return {k[3:]: gu.t(v) for k, v in d.items() if k.startswith("in/")}


def _cfg(d, **kw):
    c = dict(n_head=int(d["meta/n_head"]), eps=float(d["meta/eps"]), item="item_id")
    c.update(kw)
    return c


def _replay_mlm(d):
    ids = gu.t(d["in/item_id"])
    j2 = gu.t(d["draw/j2"])
    return O.mlm_targets_train(ids, gu.t(d["draw/bern"]), gu.t(d["draw/j1"]), lambda m: j2)


# ----------------------------------------------------------------------------- integer paths
def test_masking_int_bit_exact():
    d = gu.load("masking_int")
    ids = gu.t(d["in/item_id"])
    j2 = gu.t(d["draw/j2"])
    m, lab = O.mlm_targets_train(ids, gu.t(d["draw/bern"]), gu.t(d["draw/j1"]), lambda _: j2)
    assert torch.equal(m, gu.t(d["out/mask_schema"]))
    assert torch.equal(lab, gu.t(d["out/masked_targets"]))
    m, lab = O.mlm_targets_eval(ids, True)
    assert torch.equal(m, gu.t(d["out/eval_last_schema"])) and torch.equal(lab, gu.t(d["out/eval_last_targets"]))
    m, lab = O.mlm_targets_eval(ids, False)
    assert torch.equal(m, gu.t(d["out/eval_all_schema"])) and torch.equal(lab, gu.t(d["out/eval_all_targets"]))
    m, lab = O.mlm_targets_inference(ids)
    assert torch.equal(m, gu.t(d["out/infer_schema"])) and torch.equal(lab, gu.t(d["out/infer_targets"]))
    ids2 = gu.t(d["in/item_id_clm"])
    for tag, kw in (
        ("clm_train", dict(training=True, testing=False)),
        ("clm_train_last", dict(training=True, testing=False, train_on_last_item_seq_only=True)),
        ("clm_eval_last", dict(training=False, testing=True)),
        ("clm_eval_all", dict(training=False, testing=True, eval_on_last_item_seq_only=False)),
        ("clm_infer", dict(training=False, testing=False)),
    ):
        m, lab = O.clm_targets(ids2, **kw)
        assert torch.equal(m, gu.t(d[f"out/{tag}_schema"])), tag
        assert torch.equal(lab, gu.t(d[f"out/{tag}_targets"])), tag


def test_padding_golden():
    d = gu.load("padding")
    vals, fvals, offs = gu.t(d["in/values"]), gu.t(d["in/fvalues"]), gu.t(d["in/offsets"])
    for msl in (None, 20, 64):
        L = O.pad_inputs_length([offs], msl)
        assert torch.equal(O.pad_ragged(vals, offs, L), gu.t(d[f"out/pad_inputs_{msl}_a"]))
        assert torch.equal(O.pad_ragged(fvals, offs, L), gu.t(d[f"out/pad_inputs_{msl}_f"]))
    assert torch.equal(O.pad_ragged(vals, offs, 7), gu.t(d["out/pad_batch_7"]))
    assert torch.equal(O.pad_ragged(vals, offs, 45), gu.t(d["out/pad_batch_45"]))


def test_padding_reference_known_answers():
    """The reference's own known-answer vectors, tests/unit/utils/test_padding.py:33-151."""
    def vo(data):
        vals = [x for row in data for x in row]
        offs = np.cumsum([0] + [len(r) for r in data])
        return torch.tensor(vals, dtype=torch.int64), torch.tensor(offs, dtype=torch.int64)

    v, o = vo([[1, 2], [], [3, 4, 5]])
    assert O.pad_ragged(v, o, 7).tolist() == [[1, 2, 0, 0, 0, 0, 0], [0] * 7, [3, 4, 5, 0, 0, 0, 0]]
    v, o = vo([[1, 2], [], [3, 4, 5, 4, 7]])
    assert O.pad_ragged(v, o, 3).tolist() == [[1, 2, 0], [0, 0, 0], [3, 4, 5]]
    v, o = vo([[1, 2, 3, 4, 5], [6, 7, 8]])
    assert O.pad_ragged(v, o, O.pad_inputs_length([o])).tolist() == [[1, 2, 3, 4, 5], [6, 7, 8, 0, 0]]
    v, o = vo([[1, 2, 3, 4, 5], [6, 7, 8, 9]])
    assert O.pad_ragged(v, o, O.pad_inputs_length([o], 3)).tolist() == [[1, 2, 3], [6, 7, 8]]


def test_log_uniform_golden():
    d = gu.load("log_uniform")
    dist = O.log_uniform_dist(int(d["meta/max_id"]), int(d["meta/min_id"]))
    assert torch.equal(dist, gu.t(d["out/dist"]))
    assert torch.equal(O.unique_sampling_dist(dist, int(d["meta/n_sample"])), gu.t(d["out/unique_dist"]))


# ----------------------------------------------------------------------------- float paths
@pytest.mark.parametrize("name,masking,agg", [
    ("xlnet_mlm_item_train", "mlm", "concat"),
    ("xlnet_mlm_multi_train", "mlm", "concat"),
    ("xlnet_clm_item_train", "clm", "concat"),
    ("xlnet_mlm_context_train", "mlm", "concat"),
    ("xlnet_mlm_long_train", "mlm", "concat"),          # total_seq_length 100 (round 6: beyond one wave of the attention kernels)
    ("xlnet_clm_long_train", "clm", "concat"),
])
def test_train_forward_backward_golden(name, masking, agg):
    d = gu.load(name)
    x = _inputs(d)
    ids = x["item_id"]
    if masking == "mlm":
        m, lab = _replay_mlm(d)
    else:
        m, lab = O.clm_targets(ids, True, False)
    assert torch.equal(m, gu.t(d["out/mask_schema"])) and torch.equal(lab, gu.t(d["out/masked_targets"]))
    p = gu.oracle_params(d, requires_grad=True)
    out = O.session_forward(p, _cfg(d, masking=masking, aggregation=agg), x, m, lab, True, False)
    torch.testing.assert_close(out["inputs_embeds"], gu.t(d["out/inputs_embeds"]), **TOL)
    torch.testing.assert_close(out["hidden"], gu.t(d["out/hidden"]), **TOL)
    assert torch.equal(out["labels"], gu.t(d["out/labels"]))
    torch.testing.assert_close(out["logits"], gu.t(d["out/predictions"]), **TOL)
    torch.testing.assert_close(out["loss"], gu.t(d["out/loss"]), **TOL)
    out["loss"].backward()
    g = gu.section(d, "g/")
    item = p["tables"]["item_id"]
    torch.testing.assert_close(item.grad, g[gu.CAT + "item_id.weight"], **TOL)
    torch.testing.assert_close(p["masked_item_embedding"].grad,
                               g["heads.0.body.0._masking.masked_item_embedding"], **TOL)
    for i, lp in enumerate(p["layers"]):
        b = gu.XL + f"{i}."
        for ok, rk in (("q", "rel_attn.q"), ("k", "rel_attn.k"), ("v", "rel_attn.v"),
                       ("o", "rel_attn.o"), ("r", "rel_attn.r"), ("r_w_bias", "rel_attn.r_w_bias"),
                       ("r_r_bias", "rel_attn.r_r_bias"), ("ln_w", "rel_attn.layer_norm.weight"),
                       ("ln_b", "rel_attn.layer_norm.bias"), ("w1", "ff.layer_1.weight"),
                       ("b1", "ff.layer_1.bias"), ("w2", "ff.layer_2.weight"),
                       ("b2", "ff.layer_2.bias"), ("ff_ln_w", "ff.layer_norm.weight"),
                       ("ff_ln_b", "ff.layer_norm.bias")):
            torch.testing.assert_close(lp[ok].grad, g[b + rk], **TOL, msg=f"layer {i} {ok}")
    if p["proj"] is not None:
        torch.testing.assert_close(p["proj"][0].grad, g["heads.0.body.0.projection_module.0.0.weight"], **TOL)
    for f, tup in p["soft"].items():
        b = gu.CONT + "embedding_tables." + f
        torch.testing.assert_close(tup[0].grad, g[b + ".projection_layer.weight"], **TOL)
        torch.testing.assert_close(tup[2].grad, g[b + ".embedding_table.weight"], **TOL)
        torch.testing.assert_close(tup[3].grad, g[gu.CONT + f"post.feature_layer_norm.{f}.weight"], **TOL)
    if p["output_layer"] is not None:
        torch.testing.assert_close(p["output_layer"].grad, g[gu.TASK + "pre.module.output_layer"], **TOL)
    if p["task_proj"] is not None:
        torch.testing.assert_close(p["task_proj"][0].grad, g[gu.TASK + "task_block.0.0.weight"], **TOL)


@pytest.mark.parametrize("name", sorted(gu.PREPOST_CASES))
def test_prepost_regularisers_golden(name):
    """StochasticSwapNoise (pre) + TabularDropout / TabularLayerNorm (post) of the paper configuration,
    replaying the reference's own bernoulli / randperm / dropout draws."""
    d = gu.load(name)
    agg, post = gu.PREPOST_CASES[name]
    x, xs = gu.prepost_replay(d, O.swap_noise)
    for k in x:   # integer / value work: bit exact against every recorded augment() call
        mod = "continuous_module" if x[k].is_floating_point() else "categorical_module"
        assert torch.equal(xs[k], gu.t(d[f"out/ssn/{mod}/{k}"])), k
    assert torch.equal(xs["item_id"], gu.t(d["out/item_seq"]))
    assert not torch.equal(xs["item_id"], x["item_id"])
    # the masking runs on the swapped item ids (EmbeddingFeatures.forward stores item_seq after `pre`)
    j2 = gu.t(d["draw/j2"])
    m, lab = O.mlm_targets_train(xs["item_id"], gu.t(d["draw/bern"]), gu.t(d["draw/j1"]), lambda mm: j2)
    assert torch.equal(m, gu.t(d["out/mask_schema"])) and torch.equal(lab, gu.t(d["out/masked_targets"]))
    p = gu.oracle_params(d, requires_grad=True)
    p["post_ln"] = gu.post_ln_params(d, requires_grad=True)
    keep = {k[len("draw/post_keep/"):]: gu.t(v) for k, v in d.items() if k.startswith("draw/post_keep/")}
    assert set(keep) == set(p["tables"]) == set(p["post_ln"])
    cfg = _cfg(d, masking="mlm", aggregation=agg, post=post, post_drop_masks=keep)
    out = O.session_forward(p, cfg, xs, m, lab, True, False)
    torch.testing.assert_close(out["inputs_embeds"], gu.t(d["out/inputs_embeds"]), **TOL)
    torch.testing.assert_close(out["hidden"], gu.t(d["out/hidden"]), **TOL)
    torch.testing.assert_close(out["loss"], gu.t(d["out/loss"]), **TOL)
    out["loss"].backward()
    g = gu.section(d, "g/")
    for f, tab in p["tables"].items():
        torch.testing.assert_close(tab.grad, g[gu.CAT + f + ".weight"], **TOL)
    for f, (wk, bk) in gu.post_ln_grad_keys(d).items():
        torch.testing.assert_close(p["post_ln"][f][0].grad, g[wk], **TOL)
        torch.testing.assert_close(p["post_ln"][f][1].grad, g[bk], **TOL)


@pytest.mark.parametrize("name,params_from,masking", [
    ("xlnet_mlm_item_eval", "xlnet_mlm_item_train", "mlm"),
    ("xlnet_clm_item_eval", "xlnet_clm_item_train", "clm"),
    ("xlnet_mlm_long_eval", "xlnet_mlm_long_train", "mlm"),
])
def test_eval_golden(name, params_from, masking):
    d = gu.load(name, params_from)
    x = _inputs(d)
    ids = x["item_id"]
    m, lab = O.mlm_targets_eval(ids) if masking == "mlm" else O.clm_targets(ids, False, True)
    assert torch.equal(m, gu.t(d["out/mask_schema"])) and torch.equal(lab, gu.t(d["out/masked_targets"]))
    p = gu.oracle_params(d)
    out = O.session_forward(p, _cfg(d, masking=masking), x, m, lab, False, True)
    torch.testing.assert_close(out["hidden"], gu.t(d["out/hidden"]), **TOL)
    torch.testing.assert_close(out["logits"], gu.t(d["out/predictions"]), **TOL)
    torch.testing.assert_close(out["loss"], gu.t(d["out/loss"]), **TOL)
    assert out["logits"].shape[0] == ids.shape[0]  # one label per session (reference test_model.py:392-407)


@pytest.mark.parametrize("name,params_from,masking", [
    ("xlnet_mlm_item_infer", "xlnet_mlm_item_train", "mlm"),
    ("xlnet_clm_item_infer", "xlnet_clm_item_train", "clm"),
    ("xlnet_mlm_long_infer", "xlnet_mlm_long_train", "mlm"),       # the body runs on L + 1 = 101 positions
])
def test_inference_golden(name, params_from, masking):
    d = gu.load(name, params_from)
    x = _inputs(d)
    ids = x["item_id"]
    p = gu.oracle_params(d)
    emb = O.embedding_lookup(ids, p["tables"]["item_id"])
    if masking == "mlm":
        m, lab = O.mlm_targets_inference(ids)
        xin = O.apply_mask_mlm(emb, m, p["masked_item_embedding"], False, False)
    else:
        m, lab = O.clm_targets(ids, False, False)
        xin = O.apply_mask_clm(emb, m, p["masked_item_embedding"], False, False)
    assert torch.equal(m, gu.t(d["out/mask_schema"]))
    torch.testing.assert_close(xin, gu.t(d["out/inputs_embeds"]), **TOL)
    h = O.xlnet_model(xin, p["layers"], int(d["meta/n_head"]), float(d["meta/eps"]))
    torch.testing.assert_close(h, gu.t(d["out/hidden"]), **TOL)
    rows = O.inference_rows(h, ids, masking == "mlm")
    W = p["tables"]["item_id"] if p["output_layer"] is None else p["output_layer"]
    torch.testing.assert_close(O.head_logits(rows, W), gu.t(d["out/predictions"]), **TOL)


def test_sampled_softmax_golden():
    d = gu.load("xlnet_mlm_sum_sampled_train")
    x = _inputs(d)
    m, lab = _replay_mlm(d)
    p = gu.oracle_params(d, requires_grad=True)
    feats = {k: O.embedding_lookup(x[k], tab) for k, tab in p["tables"].items()}
    xin = O.apply_mask_mlm(O.elementwise_sum(feats), m, p["masked_item_embedding"], True, False)
    torch.testing.assert_close(xin, gu.t(d["out/inputs_embeds"]), **TOL)
    h = O.xlnet_model(xin, p["layers"], int(d["meta/n_head"]), float(d["meta/eps"]))
    xr, y = O.remove_pad_rows(h, lab)
    n = int(d["meta/max_n_samples"])
    neg = gu.t(d["draw/neg_tries"]).unique()[:n]  # prediction_task.py:843-845
    dist = gu.section(d, "p/")[gu.TASK + "pre.module.sampler.unique_sampling_dist"]
    torch.testing.assert_close(dist, O.unique_sampling_dist(O.log_uniform_dist(int(d["meta/V"]), 1), 2 * n))
    logits = O.sampled_logits(xr, y, p["tables"]["item_id"], neg, dist)
    torch.testing.assert_close(logits, gu.t(d["out/predictions"]), **TOL)
    loss = O.cross_entropy(logits, torch.zeros_like(y))
    torch.testing.assert_close(loss, gu.t(d["out/loss"]), **TOL)
    loss.backward()
    g = gu.section(d, "g/")
    torch.testing.assert_close(p["tables"]["item_id"].grad, g[gu.CAT + "item_id.weight"], **TOL)


@pytest.mark.parametrize("name,arch", [("gpt2_clm_item_train", "gpt2"), ("bert_mlm_item_train", "bert"),
                                       ("gpt2_clm_long_train", "gpt2"), ("bert_mlm_long_train", "bert")])      # total_seq_length 150
def test_gpt2_bert_block_golden(name, arch):
    """inputs_embeds -> hidden of the reference's TransformerBlock(GPT2Config / BertConfig) fixtures."""
    d = gu.load(name)
    sd = gu.section(d, "p/")
    pre = "heads.0.body.1.transformer."
    x = gu.t(d["out/inputs_embeds"])
    n = int(d["meta/n_head"])
    if arch == "gpt2":
        h = O.gpt2_model(x, O.gpt2_params_from_state(sd, pre), n, float(d["meta/eps"]))
    else:
        h = O.bert_model(x, O.bert_params_from_state(sd, pre), n, float(d["meta/eps"]))
    torch.testing.assert_close(h, gu.t(d["out/hidden"]), **TOL)
    # head on top (tied full softmax) reproduces the fixture's logits and loss
    xr, y = O.remove_pad_rows(h, gu.t(d["out/masked_targets"]))
    logits = O.head_logits(xr, sd[gu.CAT + "item_id.weight"])
    torch.testing.assert_close(logits, gu.t(d["out/predictions"]), **TOL)
    torch.testing.assert_close(O.cross_entropy(logits, y), gu.t(d["out/loss"]), **TOL)


# ----------------------------------------------------------------------------- round 3 fixtures
def test_ranking_metrics_golden():
    """the reference's NDCG / AvgPrecision / Recall / Precision / DCG @k (ranking_metric.py) per row and averaged over
    two update() calls, against (a) the oracle's general restatement and (b) the product's rank-based forms"""
    from transformers4rec_amd import ranking_metric as RM

    d = gu.load("ranking_metrics")
    scores, labels, ks = gu.t(d["in/scores"]), gu.t(d["in/labels"]), [int(k) for k in d["meta/top_ks"]]
    fns = {"ndcg_at": O.ndcg_at_k, "avg_precision_at": O.avg_precision_at_k, "recall_at": O.recall_at_k,
           "precision_at": O.precision_at_k, "dcg_at": O.dcg_at_k}
    # rank of the target: number of items scored strictly higher (no ties in random floats)
    ranks = (scores > scores.gather(1, labels[:, None])).sum(1)
    assert int((ranks == 0).sum()) >= 8
    for cls in (RM.NDCGAt, RM.AvgPrecisionAt, RM.RecallAt, RM.PrecisionAt, RM.DCGAt):
        m = cls(top_ks=ks, labels_onehot=True)
        ref_rows = gu.t(d[f"out/rows/{m.name}"])
        for j, k in enumerate(ks):
            torch.testing.assert_close(fns[m.name](scores, labels, k), ref_rows[:, j], rtol=1e-6, atol=1e-6)
        got = m.from_ranks(ranks)
        torch.testing.assert_close(got, ref_rows, rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(got.mean(0), gu.t(d[f"out/mean/{m.name}"]), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("comb", ["mean", "sum"])
def test_embedding_bag_golden(comb):
    d = gu.load(f"embedding_bag_{comb}")
    tabs = {k: gu.t(d["p/" + k]).clone().requires_grad_() for k in ("genres", "tags", "country")}
    out = {"genres": O.embedding_bag(tabs["genres"], ids=gu.t(d["in/genres"]), combiner=comb),
           "country": O.embedding_bag(tabs["country"], ids=gu.t(d["in/country"]), combiner=comb),
           "tags": O.embedding_bag(tabs["tags"], values=gu.t(d["in/tags_values"]), offsets=gu.t(d["in/tags_offsets"]),
                                   combiner=comb)}
    for k in out:
        torch.testing.assert_close(out[k], gu.t(d["out/" + k]), **TOL)
    sum((out[k] * gu.t(d["c/" + k])).sum() for k in out).backward()
    for k in out:
        torch.testing.assert_close(tabs[k].grad, gu.t(d["g/" + k]), **TOL)
    # empty bags give zero rows (torch.nn.EmbeddingBag semantics the fixture froze)
    assert float(out["tags"][0].abs().sum()) == 0.0


def _c1_check_scores(d, logits):
    cols = gu.t(d["sel/cols"])
    torch.testing.assert_close(logits[:, cols], gu.t(d["out/predictions_sel"]), **TOL)
    torch.testing.assert_close(torch.logsumexp(logits.double(), -1), gu.t(d["out/predictions_lse"]), rtol=1e-6, atol=1e-5)
    torch.testing.assert_close(logits.double().sum(-1), gu.t(d["out/predictions_rowsum"]), rtol=1e-5, atol=2e-2)
    assert torch.equal(logits.argmax(-1), gu.t(d["out/predictions_argmax"]))


def test_c1_yoochoose_golden():
    """BASELINE configs[0]: first 100 sessions of the reference's testing data, XLNet d 64 x 2 x 4 heads, MLM, tied"""
    d = gu.c1_load()
    item = "item_id/list"
    x = {item: gu.t(d["in/" + item])}
    # the padded batch IS pad_batch of the wire form the parquet holds
    assert torch.equal(O.pad_ragged(gu.t(d["in_ragged/" + item + "__values"]), gu.t(d["in_ragged/" + item + "__offsets"]),
                                    int(d["meta/L"])), x[item])
    j2 = gu.t(d["draw/j2"])
    m, lab = O.mlm_targets_train(x[item], gu.t(d["draw/bern"]), gu.t(d["draw/j1"]), lambda _: j2)
    assert torch.equal(m, gu.t(d["out/mask_schema"])) and torch.equal(lab, gu.t(d["out/masked_targets"]))
    p = gu.oracle_params(d, requires_grad=True)
    cfg = dict(n_head=int(d["meta/n_head"]), eps=float(d["meta/eps"]), item=item, masking="mlm")
    out = O.session_forward(p, cfg, x, m, lab, True, False)
    torch.testing.assert_close(out["inputs_embeds"], gu.t(d["out/inputs_embeds"]), **TOL)
    torch.testing.assert_close(out["hidden"], gu.t(d["out/hidden"]), **TOL)
    assert torch.equal(out["labels"], gu.t(d["out/labels"]))
    _c1_check_scores(d, out["logits"].detach())
    torch.testing.assert_close(out["loss"], gu.t(d["out/loss"]), **TOL)
    out["loss"].backward()
    g = gu.section(d, "g/")
    gt = p["tables"][item].grad
    key = str(d["meta/table_key"])
    torch.testing.assert_close(gt[gu.t(d["sel/cols"])], gu.t(d["gsel/" + key]), **TOL)
    torch.testing.assert_close(gt.double().sum(0), gu.t(d["gsum/" + key]), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gt.double().abs().sum(0), gu.t(d["gabs/" + key]), rtol=1e-5, atol=1e-5)
    for i, lp in enumerate(p["layers"]):
        for ok, rk in (("q", "rel_attn.q"), ("r", "rel_attn.r"), ("w1", "ff.layer_1.weight"), ("b2", "ff.layer_2.bias"),
                       ("ff_ln_w", "ff.layer_norm.weight")):
            torch.testing.assert_close(lp[ok].grad, g[gu.XL + f"{i}." + rk], **TOL)
    # eval (last item) and inference paths of the same model
    e = gu.c1_load("c1_yoochoose_eval")
    m, lab = O.mlm_targets_eval(x[item], True)
    assert torch.equal(m, gu.t(e["out/mask_schema"])) and torch.equal(lab, gu.t(e["out/masked_targets"]))
    pe = gu.oracle_params(d)
    oe = O.session_forward(pe, cfg, x, m, lab, False, True)
    _c1_check_scores(e, oe["logits"])
    torch.testing.assert_close(oe["loss"], gu.t(e["out/loss"]), **TOL)
    f = gu.c1_load("c1_yoochoose_infer")
    m, lab = O.mlm_targets_inference(x[item])
    oi = O.session_forward(pe, cfg, x, m, lab, False, False)
    _c1_check_scores(f, oi["logits"])
