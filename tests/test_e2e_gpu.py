"""End-to-end parity of the HIP path, driven through the mirrored module API
(TabularSequenceFeatures -> TransformerBlock -> NextItemPredictionTask), against
  * the fixtures produced by the unmodified reference (tests/golden, oracle/make_golden.py), and
  * the CPU oracle on larger seeded inputs, including a few optimizer steps.
Mask indices / labels: bit-exact.  Logits / loss / hidden / gradients: fp32, acceptance gate
1e-3 (north_star); the tests assert much tighter (1e-4-class) so indexing bugs cannot hide."""
import copy

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


def build_model(d, cats=(), conts=(), masking="mlm", aggregation="concat", d_output=None,
                embedding_dims=None, emb_default=None, weight_tying=True, sampled=False, max_n=100, arch="xlnet",
                context=()):
    import transformers4rec_amd as tr

    L, V = int(d["meta/L"]), int(d["meta/V"])
    schema = tr.session_schema(V - 1, L, cats, conts)
    for name, card in context:   # per-session categorical (no LIST tag, [B] ids)
        schema = schema + tr.Schema([tr.ColumnSchema(name, [tr.Tags.CATEGORICAL], tr.IntDomain(0, card))])
    kw = dict(max_sequence_length=L, masking=masking, aggregation=aggregation)
    if d_output:
        kw["d_output"] = d_output
    if conts:
        kw["continuous_soft_embeddings"] = True
    if embedding_dims:
        kw["embedding_dims"] = embedding_dims
    if emb_default:
        kw["embedding_dim_default"] = emb_default
    inputs = tr.TabularSequenceFeatures.from_schema(schema, **kw)
    kw_cfg = dict(d_model=int(d["meta/d_model"]), n_head=int(d["meta/n_head"]), n_layer=int(d["meta/n_layer"]),
                  total_seq_length=L)
    if arch == "xlnet":
        cfg = tr.XLNetConfig.build(dropout=0.0, **kw_cfg)
    elif arch == "gpt2":
        cfg = tr.GPT2Config.build(dropout=0.0, **kw_cfg)
    else:
        cfg = tr.BertConfig.build(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **kw_cfg)
    task = tr.NextItemPredictionTask(weight_tying=weight_tying, sampled_softmax=sampled, max_n_samples=max_n)
    return cfg.to_torch_model(inputs, task)


def load_reference_state(model, d):
    """Load the reference state_dict (aliases stored once) and check every parameter got a value."""
    sd = gu.section(d, "p/")
    own = model.state_dict()
    unexpected = [k for k in sd if k not in own]
    assert not unexpected, f"reference keys without a home: {unexpected}"
    loaded_ptrs = set()
    with torch.no_grad():
        for k, v in sd.items():
            assert own[k].shape == v.shape, (k, own[k].shape, v.shape)
            own[k].copy_(v)
            loaded_ptrs.add(own[k].data_ptr())
    never_used = ("r_s_bias", "seg_embed", "mask_emb", "word_embedding", "wte")  # HF params never touched on this path
    for n, p in model.named_parameters():
        assert p.data_ptr() in loaded_ptrs or any(s in n for s in never_used), f"{n} not covered by the reference state_dict"


def run_train_case(name, **build_kw):
    d = gu.load(name)
    model = build_model(d, **build_kw)
    load_reference_state(model, d)
    model.to(DEV)
    x = {k[3:]: gu.t(v).to(DEV) for k, v in d.items() if k.startswith("in/")}
    masking = model.input_features.masking
    if "draw/bern" in d:
        masking.set_draws(gu.t(d["draw/bern"]).to(DEV).to(torch.uint8), gu.t(d["draw/j1"]).to(DEV),
                          gu.t(d["draw/j2"]).to(DEV))
    cap = {}
    h0 = model.input_features.register_forward_hook(lambda m, i, o: cap.__setitem__("emb", o.detach().clone()))
    h1 = model.transformer_block.register_forward_hook(lambda m, i, o: cap.__setitem__("hid", o.detach().clone()))
    return d, model, x, cap, (h0, h1)


@pytest.mark.parametrize("name,kw", [
    ("xlnet_mlm_item_train", dict(emb_default=32)),
    ("xlnet_mlm_multi_train", dict(cats=(("category", 40), ("brand", 9)), conts=("price", "age"), d_output=32,
                                   embedding_dims={"item_id": 16, "category": 24, "brand": 8})),
    ("xlnet_clm_item_train", dict(masking="clm", emb_default=32, weight_tying=False)),
    ("xlnet_mlm_context_train", dict(context=(("country", 17),), d_output=32,
                                     embedding_dims={"item_id": 24, "country": 8})),
    ("gpt2_clm_item_train", dict(masking="clm", emb_default=32, arch="gpt2")),
    ("bert_mlm_item_train", dict(emb_default=32, arch="bert")),
    # total_seq_length 100: the general attention kernels of csrc/xlnet_attn_long.hip (round 6)
    ("xlnet_mlm_long_train", dict(emb_default=32)),
    ("xlnet_clm_long_train", dict(masking="clm", emb_default=32)),
    # total_seq_length 150, d_head 24: beyond the LDS kernels of the GPT-2 / BERT attention core (128 positions, d_head 16 / 32 / 64)
    ("gpt2_clm_long_train", dict(masking="clm", emb_default=48, arch="gpt2")),
    ("bert_mlm_long_train", dict(emb_default=48, arch="bert")),
])
def test_train_step_matches_reference(name, kw):
    d, model, x, cap, hooks = run_train_case(name, **kw)
    out = model(x, training=True)
    assert torch.equal(model.input_features.masking.mask_schema.cpu(), gu.t(d["out/mask_schema"]))
    assert torch.equal(model.input_features.masking.masked_targets.cpu(), gu.t(d["out/masked_targets"]))
    close(cap["emb"], gu.t(d["out/inputs_embeds"]))
    close(cap["hid"], gu.t(d["out/hidden"]))
    assert torch.equal(out["labels"].cpu(), gu.t(d["out/labels"]))
    close(out["predictions"], gu.t(d["out/predictions"]))
    close(out["loss"], gu.t(d["out/loss"]))
    assert abs(float(out["loss"]) - float(d["out/loss"])) < 1e-3          # the north_star gate
    assert float((out["predictions"].cpu() - gu.t(d["out/predictions"])).abs().max()) < 1e-3
    out["loss"].backward()
    g = gu.section(d, "g/")
    named = dict(model.named_parameters())
    checked = 0
    for k, ref in g.items():
        assert k in named, k
        assert named[k].grad is not None, f"no grad for {k}"
        close(named[k].grad, ref, rtol=2e-4, atol=1e-4, msg=lambda m, k=k: f"{k}: {m}")
        checked += 1
    assert checked == len(g) and checked > 10
    # parameters the reference leaves without gradient must stay untouched here too
    for k, p in named.items():
        if k not in g:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k


def test_sampled_softmax_train_matches_reference():
    name = "xlnet_mlm_sum_sampled_train"
    d, model, x, cap, _ = run_train_case(name, cats=(("category", 40),), aggregation="element-wise-sum",
                                         emb_default=32, sampled=True, max_n=20)
    # replay the reference's negative draw: monkeypatch the sampler of this task
    n = int(d["meta/max_n_samples"])
    neg = gu.t(d["draw/neg_tries"]).unique()[:n].to(DEV)
    model.prediction_task.pre.module.sampler.sample = lambda labels: neg
    out = model(x, training=True)
    close(cap["emb"], gu.t(d["out/inputs_embeds"]))
    close(out["predictions"], gu.t(d["out/predictions"]), atol=2e-4)
    close(out["loss"], gu.t(d["out/loss"]))
    assert int(out["labels"].abs().sum()) == 0
    out["loss"].backward()
    g = gu.section(d, "g/")
    named = dict(model.named_parameters())
    for k, ref in g.items():
        close(named[k].grad, ref, rtol=2e-4, atol=1e-4, msg=lambda m, k=k: f"{k}: {m}")


@pytest.mark.parametrize("name,params_from,kw", [
    ("xlnet_mlm_item_eval", "xlnet_mlm_item_train", dict(emb_default=32)),
    ("xlnet_clm_item_eval", "xlnet_clm_item_train", dict(masking="clm", emb_default=32, weight_tying=False)),
    ("xlnet_mlm_long_eval", "xlnet_mlm_long_train", dict(emb_default=32)),
])
def test_eval_matches_reference(name, params_from, kw):
    d = gu.load(name, params_from)
    model = build_model(d, **kw)
    load_reference_state(model, d)
    model.to(DEV).eval()
    x = {k[3:]: gu.t(v).to(DEV) for k, v in d.items() if k.startswith("in/")}
    with torch.no_grad():
        out = model(x, testing=True)
    assert torch.equal(model.input_features.masking.mask_schema.cpu(), gu.t(d["out/mask_schema"]))
    assert torch.equal(out["labels"].cpu(), gu.t(d["out/labels"]))
    close(out["predictions"], gu.t(d["out/predictions"]))
    close(out["loss"], gu.t(d["out/loss"]))
    assert out["predictions"].shape[0] == x["item_id"].shape[0]
    m = model.calculate_metrics(out["predictions"], out["labels"])
    ref = O.recall_at_k(gu.t(d["out/predictions"]), gu.t(d["out/labels"]), 20)
    assert torch.equal(m["recall_at_20"].cpu(), ref)
    close(m["ndcg_at_10"], O.ndcg_at_k(gu.t(d["out/predictions"]), gu.t(d["out/labels"]), 10))


@pytest.mark.parametrize("name,params_from,kw", [
    ("xlnet_mlm_item_infer", "xlnet_mlm_item_train", dict(emb_default=32)),
    ("xlnet_clm_item_infer", "xlnet_clm_item_train", dict(masking="clm", emb_default=32, weight_tying=False)),
    ("gpt2_clm_item_infer", "gpt2_clm_item_train", dict(masking="clm", emb_default=32, arch="gpt2")),
    ("bert_mlm_item_infer", "bert_mlm_item_train", dict(emb_default=32, arch="bert")),
    ("xlnet_mlm_long_infer", "xlnet_mlm_long_train", dict(emb_default=32)),       # 101 positions in the body
])
def test_inference_matches_reference(name, params_from, kw):
    d = gu.load(name, params_from)
    model = build_model(d, **kw)
    load_reference_state(model, d)
    model.to(DEV).eval()
    x = {k[3:]: gu.t(v).to(DEV) for k, v in d.items() if k.startswith("in/")}
    with torch.no_grad():
        scores = model(x)
        assert torch.equal(model.input_features.masking.mask_schema.cpu(), gu.t(d["out/mask_schema"]))
        close(scores, gu.t(d["out/predictions"]))
        model.top_k = 10
        vals, ids = model(x)
    rv, ri = torch.topk(gu.t(d["out/predictions"]), 10, dim=-1)
    assert torch.equal(ids.cpu(), ri)
    close(vals, rv)


def test_ragged_inputs_equal_padded_inputs():
    d = gu.load("xlnet_mlm_item_eval", "xlnet_mlm_item_train")
    model = build_model(d, emb_default=32)
    load_reference_state(model, d)
    model.to(DEV).eval()
    ids = gu.t(d["in/item_id"])
    lens = (ids != 0).sum(1)
    vals = ids[ids != 0]
    offs = torch.cat([torch.zeros(1, dtype=torch.int64), lens.cumsum(0)])
    with torch.no_grad():
        a = model({"item_id": ids.to(DEV)}, testing=True)
        b = model({"item_id__values": vals.to(DEV), "item_id__offsets": offs.to(DEV)}, testing=True)
    assert torch.equal(a["predictions"], b["predictions"])


def test_three_adam_steps_match_oracle():
    """C2 structure at reduced width: HIP path + FusedAdam vs oracle autograd + torch.optim.Adam,
    same injected mask draws each step."""
    import transformers4rec_amd as tr

    torch.manual_seed(0)
    B, L, V, D, n_head, n_layer = 96, 20, 2000, 64, 4, 2
    schema = tr.session_schema(V - 1, L)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="mlm",
                                                    embedding_dim_default=D)
    cfg = tr.XLNetConfig.build(D, n_head, n_layer, total_seq_length=L, dropout=0.0, initializer_range=0.05)
    model = cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=True))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(DEV)
    dense, tables = tr.flatten_model(model)
    opt = tr.FusedAdam([dense, tables], lr=1e-2)
    # oracle params from the same init (reference state_dict naming -> oracle dict)
    p = gu.oracle_params({"p/" + k: v.numpy() for k, v in sd.items()}, requires_grad=True)
    leaves = [p["tables"]["item_id"], p["masked_item_embedding"]] + [t for lp in p["layers"] for t in lp.values()]
    ref_opt = torch.optim.Adam(leaves, lr=1e-2)
    cfg_o = dict(n_head=n_head, eps=0.03, item="item_id", masking="mlm")
    g = torch.Generator().manual_seed(1)
    for step in range(3):
        data = tr.random_data_from_schema(schema, B, L, min_session_length=2, seed=100 + step)
        ids = data["item_id"]
        bern = torch.rand(B, L, generator=g) < 0.15
        lens = (ids != 0).sum(1)
        j1 = (torch.rand(B, generator=g) * lens).long()
        m1 = bern & (ids != 0)
        m1[torch.arange(B), j1] = True
        j2 = m1.float().argmax(1)
        ref_opt.zero_grad()
        m, lab = O.mlm_targets_train(ids, bern, j1, lambda mm: j2)
        ref = O.session_forward(p, cfg_o, {"item_id": ids}, m, lab, True, False)
        ref["loss"].backward()
        ref_opt.step()
        model.input_features.masking.set_draws(bern.to(DEV).to(torch.uint8), j1.to(DEV), j2.to(DEV))
        out = model({"item_id": ids.to(DEV)}, training=True)
        out["loss"].backward()
        assert torch.equal(model.input_features.masking.masked_targets.cpu(), lab)
        close(out["loss"], ref["loss"], rtol=1e-4, atol=1e-4)
        opt.step()
    close(model.input_features.item_embedding_table.weight, p["tables"]["item_id"], rtol=1e-3, atol=2e-4)
    close(model.transformer_block.transformer.layer[1].ff.layer_1.weight, p["layers"][1]["w1"], rtol=1e-3, atol=2e-4)
    # q,k,v adjacent in the flat buffer -> one batched GEMM
    ra = model.transformer_block.transformer.layer[0].rel_attn
    assert ra.k.data_ptr() == ra.q.data_ptr() + 4 * ra.q.numel()
    assert ra.v.data_ptr() == ra.k.data_ptr() + 4 * ra.k.numel()


def test_training_mode_dropout_end_to_end():
    """dropout 0.3 (the reference default) active only in module.train(); deterministic in
    (seed, step); eval mode identical to dropout 0."""
    import transformers4rec_amd as tr

    torch.manual_seed(0)
    B, L, V, D = 64, 20, 3000, 64
    schema = tr.session_schema(V - 1, L)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="mlm", embedding_dim_default=D)
    cfg = tr.XLNetConfig.build(D, 4, 2, total_seq_length=L)        # dropout=0.3 default
    assert cfg.dropout == 0.3
    model = cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=True)).to(DEV)
    x = {"item_id": tr.random_data_from_schema(schema, B, L, seed=5)["item_id"].to(DEV)}
    xl = model.transformer_block.transformer

    def run(train):
        model.train(train)
        model.input_features.masking._rng_offset = 0
        xl._drop_offset = 0
        model.zero_grad(set_to_none=True)
        out = model(x, training=True)
        out["loss"].backward()
        return out["loss"].detach().clone(), model.input_features.item_embedding_table.weight.grad.clone()

    l1, g1 = run(True)
    l2, g2 = run(True)
    assert torch.equal(l1, l2) and torch.allclose(g1, g2, rtol=1e-4, atol=1e-6)   # same (seed, step) -> same masks (atomics reorder only)
    l0, g0 = run(False)
    assert not torch.equal(l0, l1) and torch.isfinite(l1) and torch.isfinite(g1).all()
    xl._drop_offset = 0
    model.train(True)
    model.input_features.masking._rng_offset = 0
    la = model(x, training=True)["loss"]
    lb = model(x, training=True)["loss"]      # next step: new masks
    assert not torch.equal(la, lb)
    assert abs(float(l1) - float(l0)) < 0.5


def test_deferred_weight_gradient_join_gives_the_same_gradients(monkeypatch):
    """The XLNet layers hand their weight gradients to side streams and the module mirror joins them at the END of the
    backward pass (transformer.py: autograd end-of-backward callback; csrc/xlnet_layer.hip: deferred join).  At a size
    where those streams really lag (10 240 tokens, split-K weight gradients), every parameter gradient read right after
    backward() equals, bit for bit, the run that joins inside every layer call -- and a second deferred run."""
    import transformers4rec_amd as tr
    from transformers4rec_amd import transformer as T

    torch.manual_seed(0)
    B, L, V, D = 512, 20, 20001, 128     # head on csrc/head_split.hip (>= 2 GFLOP): fixed summation order there too
    schema = tr.session_schema(V - 1, L)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="mlm", embedding_dim_default=D)
    model = tr.XLNetConfig.build(D, 4, 3, total_seq_length=L).to_torch_model(
        inputs, tr.NextItemPredictionTask(weight_tying=True)).to(DEV)
    x = {"item_id": tr.random_data_from_schema(schema, B, L, seed=5)["item_id"].to(DEV)}
    xl = model.transformer_block.transformer
    model.train(True)

    def run(defer):
        monkeypatch.setattr(T, "_DEFER_JOIN", defer)
        model.input_features.masking._rng_offset = 0
        xl._drop_offset = 0
        model.zero_grad(set_to_none=True)
        model(x, training=True)["loss"].backward()
        assert not T._PENDING                      # joined (and the buffers released) before backward() returned
        return {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    joined, deferred, again = run(False), run(True), run(True)
    assert len(joined) > 40 and joined.keys() == deferred.keys()
    assert sum(".layer." in n for n in joined) == 3 * 15
    # every reduction on this path has a fixed order (head_split.hip, the sorted table scatter, the two-stage column sums
    # and split-K): ALL parameter gradients must agree bit for bit, not only the layers' own
    for n in joined:
        assert torch.equal(joined[n], deferred[n]), f"{n}: deferred join changed the gradient"
        assert torch.equal(deferred[n], again[n]), f"{n}: not reproducible"


@pytest.mark.parametrize("arch", ["gpt2", "bert"])
def test_gpt2_bert_training_mode_dropout_and_masking_rules(arch):
    import transformers4rec_amd as tr

    torch.manual_seed(0)
    B, L, V, D = 32, 20, 2000, 64
    schema = tr.session_schema(V - 1, L)
    masking = "clm" if arch == "gpt2" else "mlm"
    inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking=masking, embedding_dim_default=D)
    cfg = (tr.GPT2Config if arch == "gpt2" else tr.BertConfig).build(D, 4, 2, total_seq_length=L)
    model = cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=True)).to(DEV)
    x = {"item_id": tr.random_data_from_schema(schema, B, L, seed=5)["item_id"].to(DEV)}
    model.train()
    out = model(x, training=True)
    out["loss"].backward()
    assert torch.isfinite(out["loss"])
    g = model.input_features.item_embedding_table.weight.grad
    assert torch.isfinite(g).all() and float(g.abs().sum()) > 0
    model.eval()
    with torch.no_grad():
        a = model(x, testing=True)["loss"]
        b = model(x, testing=True)["loss"]
    assert torch.equal(a, b)
    # reference rule (tests/unit/torch/block/test_transformer.py:168-194): MLM rejected on GPT-2, CLM on BERT
    wrong = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="mlm" if arch == "gpt2" else "clm",
                                                   embedding_dim_default=D)
    with pytest.raises(ValueError, match="is not supported by"):
        tr.TransformerBlock(cfg, masking=wrong.masking)


@pytest.mark.parametrize("multi", [False, True])
def test_full_size_step_vs_oracle(multi):
    """BASELINE.json configs[1] (item-id only) and configs[2] (multi-feature) at FULL size: one
    training step (dropout 0) on the HIP path vs the CPU oracle on the same device-drawn mask."""
    import transformers4rec_amd as tr

    torch.manual_seed(0)
    torch.set_num_threads(32)
    B, L, V, D = 1024, 20, 100_000, 128
    cats = (("category", 1000), ("brand", 100), ("kind", 10)) if multi else ()
    conts = ("price", "age") if multi else ()
    schema = tr.session_schema(V, L, cats, conts)
    kw = dict(max_sequence_length=L, masking="mlm")
    if multi:
        kw.update(continuous_soft_embeddings=True, d_output=D, embedding_dims={"item_id": D}, embedding_dim_default=64)
    else:
        kw.update(embedding_dim_default=D)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, **kw)
    cfg = tr.XLNetConfig.build(D, 4, 4, total_seq_length=L, dropout=0.0)
    model = cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=True))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(DEV)
    data = tr.random_data_from_schema(schema, B, L, seed=11)
    out = model({k: v.to(DEV) for k, v in data.items()}, training=True)
    out["loss"].backward()
    m = model.input_features.masking
    mask, labels = m.mask_schema.cpu(), m.masked_targets.cpu()
    # reference invariants of the device-drawn mask at full size (test_masking.py:117-150)
    ids = data["item_id"]
    assert bool((mask.sum(1) >= 1).all()) and bool((mask.sum(1) < (ids != 0).sum(1)).all())
    assert torch.equal(labels, torch.where(mask, ids, torch.zeros_like(ids)))
    p = gu.oracle_params({"p/" + k: v.numpy() for k, v in sd.items()}, requires_grad=True)
    ref = O.session_forward(p, dict(n_head=4, eps=0.03, item="item_id", masking="mlm"), data, mask, labels, True, False)
    ref["loss"].backward()
    assert abs(float(out["loss"].detach()) - float(ref["loss"])) < 1e-4
    assert abs(float(ref["loss"]) - np.log(V + 1)) < 0.2                 # ~ln(V) at init
    assert torch.equal(out["labels"].cpu(), ref["labels"])
    assert float((out["predictions"].detach().cpu() - ref["logits"].detach()).abs().max()) < 1e-4
    g_hip = model.input_features.item_embedding_table.weight.grad.cpu()
    g_ref = p["tables"]["item_id"].grad
    close(g_hip, g_ref, rtol=1e-3, atol=2e-7)
    # checksum: softmax-CE rows sum to zero => the head's contribution to sum_v dW[v,:] vanishes;
    # what remains is the lookup scatter, identical on both sides
    close(g_hip.sum(0), g_ref.sum(0), rtol=1e-3, atol=1e-6)
    lw = model.transformer_block.transformer.layer[0].ff.layer_1.weight.grad.cpu()
    close(lw, p["layers"][0]["w1"].grad, rtol=2e-3, atol=2e-7)
    if multi:
        close(model.input_features.projection_module[0][0].weight.grad, p["proj"][0].grad, rtol=2e-3, atol=2e-7)
        close(model.input_features.continuous_module.embedding_tables["price"].embedding_table.weight.grad,
              p["soft"]["price"][2].grad, rtol=2e-3, atol=2e-6)


# ------------------------------------------------------------------------------------------
# train-time input regularisers: StochasticSwapNoise (pre), TabularDropout / TabularLayerNorm (post)
def _prepost_model(name):
    import transformers4rec_amd as tr

    d = gu.load(name)
    agg, post = gu.PREPOST_CASES[name]
    L, V, dm = int(d["meta/L"]), int(d["meta/V"]), int(d["meta/d_model"])
    concat = agg == "concat"
    cats, conts = (("category", 40), ("brand", 9)), (("price",) if concat else ())
    schema = tr.session_schema(V - 1, L, cats, conts)
    ssn = tr.StochasticSwapNoise(pad_token=0, replacement_prob=float(d["meta/ssn_p"]), schema=schema)
    post_mods = [tr.TabularDropout(op[1]) if op[0] == "dropout" else "layer-norm" for op in post]
    kw = dict(max_sequence_length=L, masking="mlm", aggregation=agg, pre=[ssn], post=post_mods)
    if concat:
        kw.update(d_output=dm, continuous_soft_embeddings=True,
                  embedding_dims={"item_id": 16, "category": 24, "brand": 8})
    else:
        kw.update(embedding_dim_default=dm)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, **kw)
    cfg = tr.XLNetConfig.build(dropout=0.0, d_model=dm, n_head=int(d["meta/n_head"]), n_layer=1, total_seq_length=L)
    model = cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=not concat))
    load_reference_state(model, d)
    model.to(DEV)
    return d, model, ssn, post


@pytest.mark.parametrize("name", sorted(gu.PREPOST_CASES))
def test_prepost_regularisers_match_reference_and_oracle(name):
    from transformers4rec_amd import features as F_, ops

    d, model, ssn, post = _prepost_model(name)
    agg = gu.PREPOST_CASES[name][0]
    x = {k[3:]: gu.t(v).to(DEV) for k, v in d.items() if k.startswith("in/")}
    draws = {}
    for k in d:
        if k.startswith("draw/ssn_bern/"):
            _, _, mod, feat = k.split("/")
            draws[(mod, feat)] = (gu.t(d[k]).to(DEV), gu.t(d[f"draw/ssn_perm/{mod}/{feat}"]).to(DEV))
    ssn.set_draws(draws)
    masking = model.input_features.masking
    masking.set_draws(gu.t(d["draw/bern"]).to(DEV).to(torch.uint8), gu.t(d["draw/j1"]).to(DEV),
                      gu.t(d["draw/j2"]).to(DEV))
    cap = {}
    model.input_features.register_forward_hook(lambda m, i, o: cap.__setitem__("emb", o.detach().clone()))
    model.transformer_block.register_forward_hook(lambda m, i, o: cap.__setitem__("hid", o.detach().clone()))
    model.train()
    out = model(x, training=True)
    # integer work: the swapped ids, the mask and the labels are the reference's, bit for bit
    cat = model.input_features.categorical_module
    assert torch.equal(cat.item_seq.cpu(), gu.t(d["out/item_seq"]))
    assert torch.equal(masking.mask_schema.cpu(), gu.t(d["out/mask_schema"]))
    assert torch.equal(masking.masked_targets.cpu(), gu.t(d["out/masked_targets"]))
    assert torch.equal(out["labels"].cpu(), gu.t(d["out/labels"]))
    # float work: the oracle with this run's own TabularDropout keep masks (Philox stream re-exported)
    _, xs = gu.prepost_replay(d, O.swap_noise)
    p = gu.oracle_params(d, requires_grad=True)
    p["post_ln"] = gu.post_ln_params(d, requires_grad=True)
    order = model.input_features._feature_order
    B, L = x["item_id"].shape
    keep = {}
    pd = float(d["meta/post_drop_p"])
    di = [i for i, op in enumerate(post) if op[0] == "dropout"][0]
    for f, tab in p["tables"].items():
        D = tab.shape[1]
        ctr = ops.dropout_ctr_hi(model.input_features._post_step, 0xFE, order.index(f) * 8 + di)
        _, m = ops.dropout(torch.ones(B * L * D, device=DEV), pd, F_.post_seed(), ctr, want_mask=True)
        keep[f] = m.view(B, L, D).cpu()
        assert 0.6 < float(keep[f].float().mean()) < 0.9
    cfg = dict(n_head=int(d["meta/n_head"]), eps=float(d["meta/eps"]), item="item_id", masking="mlm",
               aggregation=agg, post=post, post_drop_masks=keep)
    ref = O.session_forward(p, cfg, xs, gu.t(d["out/mask_schema"]), gu.t(d["out/masked_targets"]), True, False)
    close(cap["emb"], ref["inputs_embeds"])
    close(cap["hid"], ref["hidden"])
    close(out["predictions"], ref["logits"])
    close(out["loss"], ref["loss"])
    assert abs(float(out["loss"].detach()) - float(ref["loss"].detach())) < 1e-3
    out["loss"].backward()
    ref["loss"].backward()
    named = dict(model.named_parameters())
    for f, tab in p["tables"].items():
        close(named[gu.CAT + f + ".weight"].grad, tab.grad, rtol=2e-4, atol=1e-4, msg=lambda m, f=f: f"table {f}: {m}")
    for f, (wk, bk) in gu.post_ln_grad_keys(d).items():
        close(named[wk].grad, p["post_ln"][f][0].grad, rtol=2e-4, atol=1e-4)
        close(named[bk].grad, p["post_ln"][f][1].grad, rtol=2e-4, atol=1e-4)
    # eval mode: every regulariser is the identity (transformations.py:59-60; nn.Dropout)
    model.eval()
    ssn.set_draws(None)
    out_e = model(x, training=False, testing=True)
    assert torch.equal(cat.item_seq.cpu(), gu.t(d["in/item_id"]))
    keep1 = {f: torch.ones_like(k) for f, k in keep.items()}
    cfg_e = dict(cfg, post=[op if op[0] != "dropout" else ("dropout", 0.0) for op in post], post_drop_masks=keep1)
    x_cpu = {k: v.cpu() for k, v in x.items()}
    m_e, lab_e = O.mlm_targets_eval(x_cpu["item_id"])
    with torch.no_grad():
        ref_e = O.session_forward(p, cfg_e, x_cpu, m_e, lab_e, False, True)
    close(out_e["predictions"], ref_e["logits"])


def test_continuous_passthrough_columns():
    """continuous_soft_embeddings=False: ContinuousFeatures hands the values through as width-1
    columns of the concatenation (features/continuous.py:60-63)."""
    import transformers4rec_amd as tr

    L, V, B = 12, 50, 6
    schema = tr.session_schema(V, L, (("category", 7),), ("price", "age"))
    mod = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, aggregation="concat",
                                                 embedding_dims={"item_id": 8, "category": 4}).to(DEV)
    x = tr.random_data_from_schema(schema, B, L, seed=3, device=DEV)
    out = mod(x)
    names = sorted(["item_id", "category", "price", "age"])
    feats = {}
    for n in names:
        if n in mod.categorical_module.embedding_tables:
            feats[n] = O.embedding_lookup(x[n].cpu(), mod.categorical_module.embedding_tables[n].weight.detach().cpu())
        else:
            feats[n] = x[n].cpu().float().unsqueeze(-1)
    ref = O.concat_features(feats)
    assert out.shape == (B, L, 8 + 4 + 2)
    assert torch.equal(out.cpu(), ref)


# ------------------------------------------------------------------------------------------
# the other half of the BASELINE metric: Recall@20 on a held-out split after training
def _markov_sessions(n, L, V, seed, chain_seed=1234):
    """sessions from a fixed first-order Markov chain over items 1..V-1 (learnable signal)"""
    cg = torch.Generator().manual_seed(chain_seed)
    succ = torch.randint(1, V, (V, 3), generator=cg)                  # three likely successors per item
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(5, L + 1, (n,), generator=g)
    ids = torch.zeros((n, L), dtype=torch.int64)
    cur = torch.randint(1, V, (n,), generator=g)
    for t in range(L):
        ids[:, t] = cur * (t < lens)
        pick = torch.randint(0, 3, (n,), generator=g)
        noise = torch.rand(n, generator=g) < 0.1
        nxt = succ[cur, pick]
        cur = torch.where(noise, torch.randint(1, V, (n,), generator=g), nxt)
    return ids


def test_recall_at_20_after_training_matches_cpu_oracle():
    """Train the HIP path and the CPU oracle on the same Markov-chain sessions with the same initial
    weights, mask draws and Adam hyper-parameters; evaluate Recall@20 / NDCG@20 of the last item of
    held-out sessions (masking.py:461-465, ranking_metric.py:107-147, 242-280)."""
    import transformers4rec_amd as tr

    torch.manual_seed(0)
    B, L, V, D, n_head, n_layer, steps = 128, 20, 400, 64, 2, 2, 120
    schema = tr.session_schema(V - 1, L)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="mlm",
                                                    embedding_dim_default=D)
    cfg = tr.XLNetConfig.build(D, n_head, n_layer, total_seq_length=L, dropout=0.0, initializer_range=0.05)
    task = tr.NextItemPredictionTask(weight_tying=True, top_ks=(10, 20))
    model = cfg.to_torch_model(inputs, task)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(DEV)
    dense, tables = tr.flatten_model(model)
    opt = tr.FusedAdam([dense, tables], lr=3e-3)
    p = gu.oracle_params({"p/" + k: v.numpy() for k, v in sd.items()}, requires_grad=True)
    leaves = [p["tables"]["item_id"], p["masked_item_embedding"]] + [t for lp in p["layers"] for t in lp.values()]
    ref_opt = torch.optim.Adam(leaves, lr=3e-3)
    cfg_o = dict(n_head=n_head, eps=0.03, item="item_id", masking="mlm")
    train = _markov_sessions(B * 8, L, V, seed=1)
    test_ids = _markov_sessions(512, L, V, seed=2)
    g = torch.Generator().manual_seed(3)
    masking = model.input_features.masking
    for step in range(steps):
        ids = train[(step % 8) * B: (step % 8 + 1) * B]
        bern = torch.rand(B, L, generator=g) < 0.2
        lens = (ids != 0).sum(1)
        j1 = (torch.rand(B, generator=g) * lens).long()
        m1 = bern & (ids != 0)
        m1[torch.arange(B), j1] = True
        j2 = m1.float().argmax(1)
        ref_opt.zero_grad()
        m, lab = O.mlm_targets_train(ids, bern, j1, lambda mm: j2)
        ref = O.session_forward(p, cfg_o, {"item_id": ids}, m, lab, True, False)
        ref["loss"].backward()
        ref_opt.step()
        masking.set_draws(bern.to(DEV).to(torch.uint8), j1.to(DEV), j2.to(DEV))
        out = model({"item_id": ids.to(DEV)}, training=True)
        out["loss"].backward()
        opt.step()
    assert float(out["loss"]) < 0.8 * float(np.log(V))                   # it learnt something
    close(out["loss"], ref["loss"], rtol=3e-2, atol=3e-2)                 # 120 fp32 Adam steps apart
    # held-out evaluation: last item of every session
    model.eval()
    task.reset_metrics()
    with torch.no_grad():
        ev = model({"item_id": test_ids.to(DEV)}, testing=True)
        hip_metrics = task.calculate_metrics(ev["predictions"], ev["labels"])
        m_e, lab_e = O.mlm_targets_eval(test_ids)
        ref_e = O.session_forward(p, cfg_o, {"item_id": test_ids}, m_e, lab_e, False, True)
    r20_ref = float(O.recall_at_k(ref_e["logits"], ref_e["labels"], 20).mean())
    n20_ref = float(O.ndcg_at_k(ref_e["logits"], ref_e["labels"], 20).mean())
    r20 = float(hip_metrics["recall_at_20"].mean())
    n20 = float(hip_metrics["ndcg_at_20"].mean())
    agg = task.compute_metrics()
    assert abs(agg["next-item/recall_at_20"] - r20) < 1e-6
    assert r20 > 10 * 20 / V                                             # far above chance (0.05)
    assert abs(r20 - r20_ref) < 0.04 and abs(n20 - n20_ref) < 0.04, (r20, r20_ref, n20, n20_ref)
    # the metric kernel itself on the oracle's logits: exact
    mm = task.calculate_metrics(ref_e["logits"].to(DEV).contiguous(), ref_e["labels"].to(DEV))
    assert abs(float(mm["recall_at_20"].mean()) - r20_ref) < 1e-6
    assert abs(float(mm["ndcg_at_20"].mean()) - n20_ref) < 1e-5


def test_fused_eval_ranks_equal_materialised_metrics():
    """evaluate_ranks (no [N, V] scores) gives exactly the Recall/NDCG of calculate_metrics(predictions)."""
    d = gu.load("xlnet_mlm_multi_train")
    model = build_model(d, cats=(("category", 40), ("brand", 9)), conts=("price", "age"), d_output=32,
                        embedding_dims={"item_id": 16, "category": 24, "brand": 8})
    load_reference_state(model, d)
    model.to(DEV).eval()
    x = {k[3:]: gu.t(v).to(DEV) for k, v in d.items() if k.startswith("in/")}
    task = model.prediction_task
    with torch.no_grad():
        cap = {}
        h = model.transformer_block.register_forward_hook(lambda m, i, o: cap.__setitem__("hid", o))
        out = model(x, testing=True)
        h.remove()
        task.reset_metrics()
        a = task.calculate_metrics(out["predictions"], out["labels"])
        agg_a = task.compute_metrics()
        task.reset_metrics()
        b = task.evaluate_ranks(cap["hid"])
        agg_b = task.compute_metrics()
    assert torch.equal(b["labels"], out["labels"])
    for k in a:
        assert torch.equal(a[k], b["metrics"][k]), k
    assert agg_a == agg_b and len(agg_a) == 6       # NDCG, AvgPrecision, Recall @ 10, 20 (the reference default set)


# ------------------------------------------------------------------------------------------
# non-materialising head and the large configurations (items 1a, 1c)
@pytest.mark.parametrize("name,kw", [
    ("xlnet_mlm_item_train", dict(emb_default=32)),
    ("xlnet_mlm_multi_train", dict(cats=(("category", 40), ("brand", 9)), conts=("price", "age"), d_output=32,
                                   embedding_dims={"item_id": 16, "category": 24, "brand": 8})),
    ("xlnet_clm_item_train", dict(masking="clm", emb_default=32, weight_tying=False)),
])
def test_fused_head_matches_reference_and_materialised_path(name, kw, monkeypatch):
    """head_mode='fused' (no [N, V] logits, lazy predictions) against the reference fixture and against the
    materialised path: loss <= 1e-5, every gradient, and predictions once somebody asks for them."""
    from transformers4rec_amd.prediction_task import LazyPredictions

    res = {}
    for mode in ("materialize", "fused"):
        d, model, x, cap, hooks = run_train_case(name, **kw)
        model.prediction_task.head_mode = mode
        out = model(x, training=True)
        out["loss"].backward()
        res[mode] = (out, {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    (om, gm), (of, gf) = res["materialize"], res["fused"]
    assert isinstance(of["predictions"], LazyPredictions) and not of["predictions"].is_materialized
    assert torch.is_tensor(om["predictions"])
    assert abs(float(of["loss"].detach()) - float(om["loss"].detach())) < 1e-5
    assert abs(float(of["loss"].detach()) - float(d["out/loss"])) < 1e-4
    assert torch.equal(of["labels"], om["labels"])
    assert sorted(gm) == sorted(gf)
    for n in gm:
        close(gf[n], gm[n], rtol=1e-4, atol=2e-7, msg=lambda m, n=n: f"{n}: {m}")
    ref_g = gu.section(d, "g/")
    for n, p in dict(gf).items():
        if n in ref_g:
            close(p, ref_g[n], rtol=2e-3, atol=2e-6, msg=lambda m, n=n: f"{n} vs reference: {m}")
    # lazy predictions: shape known without computing; any tensor use computes the same logits
    lp = of["predictions"]
    assert tuple(lp.shape) == tuple(om["predictions"].shape) and not lp.is_materialized
    close(torch.softmax(lp, -1), torch.softmax(om["predictions"], -1), rtol=0, atol=1e-6)
    assert lp.is_materialized
    close(lp.materialize(), gu.t(d["out/predictions"]), rtol=0, atol=1e-4)
    m = model.calculate_metrics(lp, of["labels"])
    assert set(m) >= {"recall_at_10", "ndcg_at_20"}


def test_fused_head_allocates_no_logits_and_auto_mode(monkeypatch):
    """no [N, V] allocation on the fused path (torch.cuda.max_memory_allocated), 'auto' picks it by size"""
    import transformers4rec_amd as tr

    B, L, V, D = 512, 20, 200_000, 64
    schema = tr.session_schema(V, L)
    peaks = {}
    for mode in ("fused", "materialize"):
        torch.manual_seed(0)
        inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="clm", embedding_dim_default=D)
        cfg = tr.XLNetConfig.build(D, 2, 1, total_seq_length=L, dropout=0.0)
        model = cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=True, head_mode=mode)).to(DEV)
        batch = tr.random_data_from_schema(schema, B, L, seed=2, device=DEV)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        out = model(batch, training=True)
        out["loss"].backward()
        torch.cuda.synchronize()
        N = out["labels"].numel()
        peaks[mode] = (torch.cuda.max_memory_allocated() - base, 4 * N * V, float(out["loss"].detach()))
        del model, out
    (pf, logits_bytes, lf), (pm, _, lm) = peaks["fused"], peaks["materialize"]
    assert logits_bytes > 2e9                                   # the test is about a tensor worth avoiding
    assert pm > logits_bytes                                    # the materialised path holds it ...
    assert pf < 0.25 * logits_bytes, (pf, logits_bytes)         # ... the fused one never does
    assert abs(lf - lm) < 1e-5
    t = tr.NextItemPredictionTask()
    assert t.resolve_head_mode(2765, 100_001) == "materialize" and t.resolve_head_mode(15_360, 10_000_001) == "fused"
    monkeypatch.setenv("T4R_HEAD_MODE", "fused")
    assert t.resolve_head_mode(10, 10) == "fused"


def _gpt2_oracle_params(sd):
    return O.gpt2_params_from_state({k: v.clone().requires_grad_() for k, v in sd.items()
                                     if k.startswith("heads.0.body.1.transformer.")}, "heads.0.body.1.transformer.")


def test_c4_full_size_step_vs_oracle():
    """BASELINE.json configs[3] at FULL size on one GPU's share: GPT-2 d_model 256, 6 layers, 1 M items, seq 50,
    batch 1024, causal LM, sampled softmax (100 negatives, log-uniform), tied weights.  One training step (dropout 0)
    on the HIP path vs the CPU oracle with the same negatives: loss, labels, table gradient, a layer gradient."""
    import transformers4rec_amd as tr

    torch.manual_seed(0)
    torch.set_num_threads(32)
    B, L, V, D, NH, NL = 1024, 50, 1_000_000, 256, 4, 6
    schema = tr.session_schema(V, L)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="clm", embedding_dim_default=D)
    cfg = tr.GPT2Config.build(D, NH, NL, total_seq_length=L, dropout=0.0)
    model = cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=True, sampled_softmax=True, max_n_samples=100))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(DEV)
    data = tr.random_data_from_schema(schema, B, L, seed=21)
    sampler = model.prediction_task.pre.module.sampler
    torch.manual_seed(3)
    neg = sampler.sample(torch.ones(4, dtype=torch.long, device=DEV))
    sampler.sample = lambda labels: neg
    out = model({k: v.to(DEV) for k, v in data.items()}, training=True)
    out["loss"].backward()
    ids = data["item_id"]
    mask, labels = O.clm_targets(ids, True, False)
    m = model.input_features.masking
    assert torch.equal(m.mask_schema.cpu(), mask) and torch.equal(m.masked_targets.cpu(), labels)   # integer work: exact
    # oracle: lookup -> CLM input masking -> GPT-2 -> label rows -> sampled logits -> CE
    table = sd["heads.0.body.0.to_merge.categorical_module.embedding_tables.item_id.weight"].clone().requires_grad_()
    memb = sd["heads.0.body.0._masking.masked_item_embedding"].clone().requires_grad_()
    P = _gpt2_oracle_params(sd)
    x = O.apply_mask_clm(O.embedding_lookup(ids, table), mask, memb, True, False)
    h = O.gpt2_model(x, P, NH)
    xr, y = O.remove_pad_rows(h, labels)
    dist = O.unique_sampling_dist(O.log_uniform_dist(V + 1, 1), 200)
    close(sampler.unique_sampling_dist, dist, rtol=1e-6, atol=1e-12)
    logits = O.sampled_logits(xr, y, table, neg.cpu(), dist, 1.0)
    loss = O.cross_entropy(logits, torch.zeros_like(y))
    loss.backward()
    assert out["labels"].numel() == y.numel() and y.numel() > 20_000
    assert abs(float(out["loss"].detach()) - float(loss)) < 1e-4
    assert float((out["predictions"].detach().cpu() - logits.detach()).abs().max()) < 1e-3
    g_hip = model.input_features.item_embedding_table.weight.grad.cpu()
    close(g_hip, table.grad, rtol=1e-3, atol=2e-7)
    close(g_hip.sum(0), table.grad.sum(0), rtol=1e-3, atol=1e-6)
    blk = model.transformer_block.transformer.h[0]
    close(blk.mlp.c_fc.weight.grad, P["blocks"][0]["c_fc_w"].grad, rtol=2e-3, atol=2e-7)
    close(model.transformer_block.transformer.wpe.weight.grad[:L], P["wpe"].grad[:L], rtol=2e-3, atol=2e-7)
    close(m.masked_item_embedding.grad, memb.grad, rtol=2e-3, atol=2e-7)


def _bert_c5_model(tr, V, L, D, NH, NL, device=None):
    schema = tr.session_schema(V, L)
    ctx = torch.device(device) if device else torch.device("cpu")
    with ctx:
        inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="mlm", embedding_dim_default=D)
        cfg = tr.BertConfig.build(D, NH, NL, total_seq_length=L, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        model = cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=True, head_mode="fused"))
    return model, schema


def test_c5_reduced_vocab_parity_vs_oracle():
    """BASELINE.json configs[4] body (BERT d_model 512, 12 layers, 8 heads, seq 100, MLM, tied full softmax) at a
    CPU-checkable size (1 M items, batch 64): one training step on the non-materialising head vs the oracle."""
    import transformers4rec_amd as tr

    torch.manual_seed(0)
    torch.set_num_threads(32)
    B, L, V, D, NH, NL = 64, 100, 1_000_000, 512, 8, 12
    model, schema = _bert_c5_model(tr, V, L, D, NH, NL)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(DEV)
    data = tr.random_data_from_schema(schema, B, L, seed=5)
    out = model({k: v.to(DEV) for k, v in data.items()}, training=True)
    out["loss"].backward()
    m = model.input_features.masking
    mask, labels = m.mask_schema.cpu(), m.masked_targets.cpu()
    ids = data["item_id"]
    assert torch.equal(labels, torch.where(mask, ids, torch.zeros_like(ids)))
    table = sd["heads.0.body.0.to_merge.categorical_module.embedding_tables.item_id.weight"].clone().requires_grad_()
    memb = sd["heads.0.body.0._masking.masked_item_embedding"].clone().requires_grad_()
    pre = "heads.0.body.1.transformer."
    P = O.bert_params_from_state({k: v.clone().requires_grad_() for k, v in sd.items() if k.startswith(pre)}, pre)
    x = O.apply_mask_mlm(O.embedding_lookup(ids, table), mask, memb, True, False)
    h = O.bert_model(x, P, NH, 0.03)
    xr, y = O.remove_pad_rows(h, labels)
    loss = O.cross_entropy(O.head_logits(xr, table, 1.0), y)
    loss.backward()
    assert torch.equal(out["labels"].cpu(), y)
    assert abs(float(out["loss"].detach()) - float(loss)) < 1e-4
    g_hip = model.input_features.item_embedding_table.weight.grad.cpu()
    close(g_hip, table.grad, rtol=1e-3, atol=2e-7)
    close(g_hip.sum(0), table.grad.sum(0), rtol=1e-3, atol=2e-6)
    lay = model.transformer_block.transformer.encoder.layer[0]
    close(m.masked_item_embedding.grad, memb.grad, rtol=2e-3, atol=2e-7)
    close(lay.intermediate.dense.weight.grad, P["layers"][0]["i_w"].grad, rtol=2e-3, atol=2e-7)
    close(model.transformer_block.transformer.embeddings.position_embeddings.weight.grad[:L], P["pos"].grad[:L],
          rtol=2e-3, atol=2e-7)


def test_c5_full_vocabulary_step_runs():
    """C5 at its vocabulary: 10 M items x 512 (20 GB fp32 table), 12-layer BERT, seq 100 -- one fwd+bwd step on the
    non-materialising head (the [N, V] logits would be 4 * N * 10^7 bytes: 150 GB at this batch of 256).  Checks
    what is size-independent: loss ~ ln V at init, softmax rows sum to one => the head's table gradient columns
    sum to zero, i.e. sum_v dW[v, :] equals the lookup scatter alone; peak memory far below the logits."""
    import transformers4rec_amd as tr

    torch.manual_seed(0)
    B, L, V, D, NH, NL = 256, 100, 10_000_000, 512, 8, 12
    model, schema = _bert_c5_model(tr, V, L, D, NH, NL, device=DEV)
    model.train()
    batch = tr.random_data_from_schema(schema, B, L, seed=9, device=DEV)
    cap = {}
    def grab(mod, i, o):        # (a forward hook's return value would replace the output)
        o.register_hook(lambda gr: cap.__setitem__("d_emb", gr.detach().clone()))

    model.input_features.register_forward_hook(grab)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    out = model(batch, training=True)
    out["loss"].backward()
    torch.cuda.synchronize()
    N = out["labels"].numel()
    peak = torch.cuda.max_memory_allocated() - base
    # at init the logits are ~N(0, s^2) with s = |h| * 0.05 ~ 1.1 (LayerNorm-ed hidden of width 512): loss ~ ln V + s^2/2
    loss0 = float(out["loss"].detach())
    assert N > 2000 and 0.0 < loss0 - np.log(V + 1) < 1.2, loss0
    table = model.input_features.item_embedding_table.weight
    assert peak < 4.0 * table.numel() * 4 and peak < 0.5 * 4.0 * N * V, (peak, 4.0 * N * V)
    g = table.grad
    assert torch.isfinite(g).all()
    # checksum of checksums: rows of (softmax - onehot) sum to zero, so the head's dW columns sum to zero and
    # sum_v g[v, :] is the lookup scatter alone = sum of d loss / d inputs_embeds over the positions whose
    # embedding reached the body (non-pad, not replaced by the [MASK] vector)
    ids, mask = batch["item_id"], model.input_features.masking.mask_schema
    want = cap["d_emb"][(ids != 0) & ~mask].double().sum(0)
    got = g.double().sum(0)
    assert float(g.abs().max()) > 0
    assert float((got - want).abs().max()) < 1e-2 * float(want.abs().max()) + 1e-6, (got[:4], want[:4])


@pytest.mark.parametrize("name,kw", [
    ("xlnet_mlm_multi_train", dict(cats=(("category", 40), ("brand", 9)), conts=("price", "age"), d_output=32,
                                   embedding_dims={"item_id": 16, "category": 24, "brand": 8})),
    ("xlnet_mlm_context_train", dict(context=(("country", 17),), d_output=32,
                                     embedding_dims={"item_id": 24, "country": 8})),
    ("xlnet_mlm_sum_sampled_train", dict(cats=(("category", 40),), aggregation="element-wise-sum", emb_default=32,
                                         sampled=True, max_n=20)),
])
def test_row_sparse_sink_gives_the_same_table_gradients(name, kw):
    """the data-parallel row-sparse path on one process: table gradients collected as (ids, rows) by a
    SparseRowExchange and applied by GradReducer.reduce_all == the direct scatter, and == the reference"""
    import transformers4rec_amd as tr

    grads = {}
    for use_sink in (False, True):
        d, model, x, cap, hooks = run_train_case(name, **kw)
        if "draw/neg_tries" in d:
            neg = gu.t(d["draw/neg_tries"]).unique()[: int(d["meta/max_n_samples"])].to(DEV)
            model.prediction_task.pre.module.sampler.sample = lambda labels, neg=neg: neg
        tabs = [p for n, p in model.named_parameters() if ".embedding_tables." in n and "continuous_module" not in n]
        red = None
        if use_sink:
            sink = tr.SparseRowExchange().attach(*tabs)
            red = tr.GradReducer(torch.zeros(1, device=DEV), None, sparse=sink)
        out = model(x, training=True)
        out["loss"].backward()
        if use_sink:
            assert len(sink._pending) >= len(tabs) - 1
            red.reduce_all()
            assert not sink._pending
        grads[use_sink] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    assert sorted(grads[False]) == sorted(grads[True])
    for n in grads[False]:
        close(grads[True][n], grads[False][n], rtol=1e-5, atol=1e-7, msg=lambda m, n=n: f"{n}: {m}")
    ref_g = gu.section(d, "g/")
    for n, g in grads[True].items():
        if n in ref_g and ".embedding_tables." in n:
            close(g, ref_g[n], rtol=2e-4, atol=1e-4, msg=lambda m, n=n: f"{n} vs reference: {m}")


# ------------------------------------------------------------------------------------------
# precision modes of the dense contractions
@pytest.mark.parametrize("mode", ["fp32_bf16x3", "auto"])
def test_fp32_accurate_split_mode_matches_reference(mode):
    """the fp32-accurate form on the bf16 matrix cores (exact 3-way split) keeps the reference parity of the fp32
    path: fixture loss / logits / every gradient at the SAME tolerances as test_train_step_matches_reference"""
    from transformers4rec_amd import ops

    name, kw = "xlnet_mlm_multi_train", dict(cats=(("category", 40), ("brand", 9)), conts=("price", "age"), d_output=32,
                                             embedding_dims={"item_id": 16, "category": 24, "brand": 8})
    d, model, x, cap, hooks = run_train_case(name, **kw)
    with ops.precision(mode):
        out = model(x, training=True)
        out["loss"].backward()
    close(cap["hid"], gu.t(d["out/hidden"]))
    close(out["predictions"], gu.t(d["out/predictions"]))
    assert abs(float(out["loss"].detach()) - float(d["out/loss"])) < 1e-4
    named = dict(model.named_parameters())
    for k, ref in gu.section(d, "g/").items():
        close(named[k].grad, ref, rtol=2e-4, atol=1e-4, msg=lambda m, k=k: f"{k}: {m}")


@pytest.mark.parametrize("smooth,temp", [(0.0, 1.0), (0.1, 0.7)])
def test_head_split_path_is_taken_and_agrees_with_the_general_gemm(monkeypatch, smooth, temp):
    """the d_model <= 128 head (csrc/head_split.hip) replaces the general GEMM for >= 2 GFLOP products in the
    fp32-accurate modes: same loss, predictions and gradients as the general path on the same model and batch"""
    import transformers4rec_amd as tr
    from transformers4rec_amd import ops, prediction_task as pt

    torch.manual_seed(0)
    V, L, D, B = 30_000, 20, 128, 128
    schema = tr.session_schema(V, L)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=L, masking="mlm", embedding_dim_default=D)
    task = tr.NextItemPredictionTask(weight_tying=True, softmax_temperature=temp,
                                     loss=torch.nn.CrossEntropyLoss(label_smoothing=smooth))
    model = tr.XLNetConfig.build(D, 4, 2, total_seq_length=L).to_torch_model(inputs, task).to("cuda")
    x = tr.random_data_from_schema(schema, B, L, seed=1, device="cuda")
    taken = []
    # either form of csrc/head_split.hip's forward: the one-pass logits + CE + d X (round 5, default) or logits + CE
    real, real_dx = ops.head_split_logits_ce, ops.head_split_logits_ce_dx
    monkeypatch.setattr(ops, "head_split_logits_ce", lambda *a, **k: (taken.append(1), real(*a, **k))[1])
    monkeypatch.setattr(ops, "head_split_logits_ce_dx", lambda *a, **k: (taken.append(1), real_dx(*a, **k))[1])
    from transformers4rec_amd.rng import get_rng_state, set_rng_state

    res = {}
    model.train()
    state = get_rng_state(model)
    for on in (True, False):
        monkeypatch.setattr(pt, "_HEAD_SPLIT", on)
        model.zero_grad(set_to_none=True)
        set_rng_state(model, state)                             # the same MLM mask and dropout masks both times
        out = model(x, training=True)
        out["loss"].backward()
        torch.cuda.synchronize()
        res[on] = (float(out["loss"].detach()), out["predictions"].detach().clone(),
                   {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    assert len(taken) == 1
    assert abs(res[True][0] - res[False][0]) < 1e-5
    close(res[True][1], res[False][1], rtol=1e-5, atol=1e-5)
    for n, g in res[True][2].items():
        close(g, res[False][2][n], rtol=2e-4, atol=1e-6, msg=lambda m, n=n: f"{n}: {m}")


@pytest.mark.parametrize("mode,dtype", [("bf16", torch.bfloat16), ("fp16", torch.float16)])
def test_mixed_precision_mode(mode, dtype):
    """C5's precision mode (reference: HF Trainer fp16=True -> autocast, trainer.py:363-367; master weights fp32):
    half-precision operands in every dense contraction, fp32 accumulation / LayerNorm / softmax / CE.
    Reported separately from the fp32 gate: compared with (a) the fp32 HIP path and (b) the CPU oracle
    under torch.autocast -- the reference's AMP semantics -- at half-precision tolerances."""
    import transformers4rec_amd as tr
    from transformers4rec_amd import ops

    name, kw = "xlnet_mlm_item_train", dict(emb_default=32)
    res = {}
    for m in ("fp32", mode):
        d, model, x, cap, hooks = run_train_case(name, **kw)
        with ops.precision(m):
            out = model(x, training=True)
            out["loss"].backward()
        res[m] = (float(out["loss"].detach()), out["predictions"].detach().cpu(), cap["hid"].cpu(),
                  {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None})
    l32, p32, h32, g32 = res["fp32"]
    lh, ph, hh, gh = res[mode]
    eps = 2.0 ** -8 if mode == "bf16" else 2.0 ** -11
    assert lh != l32, "the mode did not change the arithmetic"
    assert abs(lh - l32) < 40 * eps * max(1.0, abs(l32)), (lh, l32)
    assert float((ph - p32).abs().max()) < 60 * eps * float(p32.abs().max())
    assert float((hh - h32).abs().max()) < 60 * eps * float(h32.abs().max())
    for n in g32:
        a, b = gh[n], g32[n]
        assert float((a - b).abs().max()) <= 100 * eps * float(b.abs().max()) + 1e-7, n
    # (b) the reference's AMP semantics on the CPU: the oracle under autocast
    mk = model.input_features.masking
    p = gu.oracle_params(d, requires_grad=False)
    ids = {k[3:]: gu.t(v) for k, v in d.items() if k.startswith("in/")}
    with torch.autocast(device_type="cpu", dtype=dtype):
        ref = O.session_forward(p, dict(n_head=int(d["meta/n_head"]), eps=0.03, item="item_id", masking="mlm"), ids,
                                gu.t(d["out/mask_schema"]), gu.t(d["out/masked_targets"]), True, False)
    assert abs(lh - float(ref["loss"])) < 60 * eps * max(1.0, abs(l32)), (lh, float(ref["loss"]))


def test_mask_padding_option_end_to_end():
    """TransformerBlock(mask_padding=True): train step vs the oracle with the same key mask; the default (False)
    stays the reference's unmasked attention (fixture parity is what every other test checks)."""
    import transformers4rec_amd as tr

    d = gu.load("xlnet_mlm_item_train")
    res = {}
    for mp in (False, True):
        model = build_model(d, emb_default=32)
        load_reference_state(model, d)
        model.transformer_block.mask_padding = mp
        model.to(DEV)
        x = {k[3:]: gu.t(v).to(DEV) for k, v in d.items() if k.startswith("in/")}
        model.input_features.masking.set_draws(gu.t(d["draw/bern"]).to(DEV).to(torch.uint8), gu.t(d["draw/j1"]).to(DEV),
                                               gu.t(d["draw/j2"]).to(DEV))
        out = model(x, training=True)
        out["loss"].backward()
        res[mp] = (out, model)
    p = gu.oracle_params(d, requires_grad=True)
    ids = gu.t(d["in/item_id"])
    key_len = (ids != 0).sum(1).to(torch.int32)
    mask, labels = gu.t(d["out/mask_schema"]), gu.t(d["out/masked_targets"])
    xe = O.apply_mask_mlm(O.embedding_lookup(ids, p["tables"]["item_id"]), mask, p["masked_item_embedding"], True, False)
    h = O.xlnet_model(xe, p["layers"], int(d["meta/n_head"]), 0.03, key_len=key_len)
    xr, y = O.remove_pad_rows(h, labels)
    loss = O.cross_entropy(O.head_logits(xr, p["tables"]["item_id"], 1.0), y)
    loss.backward()
    out_t, model_t = res[True]
    assert abs(float(out_t["loss"].detach()) - float(loss)) < 1e-4
    close(model_t.input_features.item_embedding_table.weight.grad, p["tables"]["item_id"].grad, rtol=2e-4, atol=1e-4)
    close(model_t.transformer_block.transformer.layer[0].rel_attn.q.grad, p["layers"][0]["q"].grad, rtol=2e-4, atol=1e-4)
    assert abs(float(res[False][0]["loss"].detach()) - float(d["out/loss"])) < 1e-4       # default: the reference
    assert abs(float(out_t["loss"].detach()) - float(res[False][0]["loss"].detach())) > 1e-6


@pytest.mark.parametrize("arch,name,masking", [("gpt2", "gpt2_clm_item_train", "clm"), ("bert", "bert_mlm_item_train", "mlm")])
def test_mask_padding_option_gpt2_bert(arch, name, masking):
    """mask_padding=True on the GPT-2 / BERT bodies: hidden states of the valid positions vs the oracle with the key mask"""
    d = gu.load(name)
    model = build_model(d, masking=masking, emb_default=32, arch=arch)
    load_reference_state(model, d)
    model.transformer_block.mask_padding = True
    model.to(DEV).eval()
    x = {k[3:]: gu.t(v).to(DEV) for k, v in d.items() if k.startswith("in/")}
    cap = {}
    model.input_features.register_forward_hook(lambda m, i, o: cap.__setitem__("emb", o.detach().cpu()))
    model.transformer_block.register_forward_hook(lambda m, i, o: cap.__setitem__("hid", o.detach().cpu()))
    with torch.no_grad():
        model(x, testing=True)
    ids = gu.t(d["in/item_id"])
    key_len = (ids != 0).sum(1).to(torch.int32)
    sd = {k: gu.t(v) for k, v in d.items() if k.startswith("p/heads.0.body.1.transformer.")}
    pre = "p/heads.0.body.1.transformer."
    nh = int(d["meta/n_head"])
    if arch == "gpt2":
        ref = O.gpt2_model(cap["emb"], O.gpt2_params_from_state(sd, pre), nh, 1e-5, key_len=key_len)
        unm = O.gpt2_model(cap["emb"], O.gpt2_params_from_state(sd, pre), nh, 1e-5)
    else:
        ref = O.bert_model(cap["emb"], O.bert_params_from_state(sd, pre), nh, 0.03, key_len=key_len)
        unm = O.bert_model(cap["emb"], O.bert_params_from_state(sd, pre), nh, 0.03)
    valid = ids != 0
    close(cap["hid"][valid], ref[valid], rtol=1e-4, atol=5e-5)
    if arch == "bert":          # (a causal body never lets a valid query see a padded key: the mask changes nothing there)
        assert float((ref[valid] - unm[valid]).abs().max()) > 1e-5
