"""GPU half of the drop-in adapter (transformers4rec_amd/dropin.py): forward / backward THROUGH the
shadow modules.  The reference source tree does not exist on the GPU box, so the adapter is driven
with this package's own mirror classes standing in for the reference namespace (they carry the
reference's module tree and state_dict names, tests/test_host_contract.py): a model whose three
hot-path modules were class-swapped to the Hip* subclasses must give the same loss / predictions /
parameter gradients as the same model called directly, with the gradients landing in the SOURCE
model's parameters, and must not register anything new.  The structural half against the real
reference (isinstance gates, to_torch_model, state_dict round trip) is tests/test_dropin_cpu.py.
"""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = {
    "xlnet_mlm_tied": dict(),
    "xlnet_mlm_multi_proj": dict(cats=(("category", 40), ("brand", 9)), conts=("price", "age"), d_output=32,
                                 embedding_dims={"item_id": 16, "category": 24, "brand": 8}),
    "xlnet_clm_untied": dict(masking="clm", weight_tying=False),
    "xlnet_mlm_sum_sampled": dict(cats=(("category", 40),), aggregation="element-wise-sum", sampled=True),
    "gpt2_clm": dict(arch="gpt2", masking="clm"),
    "bert_mlm": dict(arch="bert"),
    "xlnet_mlm_task_block": dict(d_output=32, embedding_dims={"item_id": 24}),
}


def _build(tr, arch="xlnet", masking="mlm", cats=(), conts=(), d_output=None, weight_tying=True, sampled=False,
           aggregation="concat", **fkw):
    V, L, d = 300, 20, 32
    schema = tr.session_schema(V, L, cats, conts)
    kw = dict(max_sequence_length=L, masking=masking, aggregation=aggregation, embedding_dim_default=32)
    if conts:
        kw["continuous_soft_embeddings"] = True
    if d_output:
        kw["d_output"] = d_output
    kw.update(fkw)
    torch.manual_seed(0)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, **kw)
    cfgc = {"xlnet": tr.XLNetConfig, "gpt2": tr.GPT2Config, "bert": tr.BertConfig}[arch]
    ck = dict(d_model=d, n_head=2, n_layer=2, total_seq_length=L)
    if arch != "bert":
        ck["dropout"] = 0.0
    else:
        ck.update(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    cfg = cfgc.build(**ck)
    task = tr.NextItemPredictionTask(weight_tying=weight_tying, sampled_softmax=sampled, max_n_samples=20)
    return cfg.to_torch_model(inputs, task), schema


@pytest.mark.parametrize("name", sorted(CASES))
def test_swapped_model_equals_direct_call(name):
    import transformers4rec_amd as tr
    from transformers4rec_amd import dropin

    dev = torch.device("cuda", 0)
    direct, schema = _build(tr, **CASES[name])
    swapped, _ = _build(tr, **CASES[name])
    swapped.load_state_dict(direct.state_dict())
    direct.to(dev).train()
    swapped.to(dev).train()
    ns = types.SimpleNamespace(TabularSequenceFeatures=tr.TabularSequenceFeatures,
                               TransformerBlock=tr.TransformerBlock,
                               NextItemPredictionTask=tr.NextItemPredictionTask)
    keys = list(swapped.state_dict().keys())
    dropin.convert_model(swapped, ns)
    feats, block, task = swapped.input_features, swapped.transformer_block, swapped.prediction_task
    assert all(getattr(m, "_t4r_hip", False) for m in (feats, block, task))
    assert isinstance(feats, tr.TabularSequenceFeatures) and isinstance(task, tr.NextItemPredictionTask)

    batch = tr.random_data_from_schema(schema, 16, 20, seed=3, device=dev)
    for m in (direct.input_features.masking, feats.hip_shadow().masking):
        m.seed, m._rng_offset = 77, 0
    outs = []
    for model in (direct, swapped):
        torch.manual_seed(5)          # the sampled-softmax negatives come from torch.multinomial
        out = model(batch, training=True)
        out["loss"].backward()
        outs.append(out)
    assert list(swapped.state_dict().keys()) == keys, "the shadows registered something"
    a, b = outs
    assert torch.equal(a["labels"], b["labels"])
    assert abs(float(a["loss"]) - float(b["loss"])) < 1e-6
    assert torch.allclose(a["predictions"], b["predictions"], rtol=0, atol=1e-6)
    # the reference-side stateful attributes are set (masking.py:148-152, embedding.py:242-245)
    assert feats.masking.masked_targets is feats.hip_shadow().masking.masked_targets
    assert torch.equal(feats.masking.mask_schema, direct.input_features.masking.mask_schema)
    assert torch.equal(feats.categorical_module.item_seq, batch["item_id"])
    # gradients landed in the SOURCE model's parameters
    gd = {n: p.grad for n, p in direct.named_parameters() if p.grad is not None}
    gs = {n: p.grad for n, p in swapped.named_parameters() if p.grad is not None}
    assert gd and sorted(gd) == sorted(gs)
    for n in gd:      # split-K / scatter atomics: the summation order differs from launch to launch
        assert torch.allclose(gd[n], gs[n], rtol=1e-4, atol=1e-7), (n, float((gd[n] - gs[n]).abs().max()))

    # evaluation and inference go through the shadows as well
    direct.eval()
    swapped.eval()
    ea, eb = direct(batch, testing=True), swapped(batch, testing=True)
    assert torch.allclose(ea["predictions"], eb["predictions"], rtol=0, atol=1e-6) and torch.equal(ea["labels"], eb["labels"])
    ia, ib = direct(batch), swapped(batch)
    assert torch.allclose(ia, ib, rtol=0, atol=1e-6)


def test_task_without_hip_features_raises():
    import transformers4rec_amd as tr
    from transformers4rec_amd import dropin

    dev = torch.device("cuda", 0)
    model, schema = _build(tr)
    model.to(dev)
    ns = types.SimpleNamespace(TabularSequenceFeatures=type("Other", (), {}), TransformerBlock=type("Other2", (), {}),
                               NextItemPredictionTask=tr.NextItemPredictionTask)
    dropin.convert_model(model, ns)       # only the task is swapped
    batch = tr.random_data_from_schema(schema, 8, 20, seed=1, device=dev)
    with pytest.raises(RuntimeError, match="label compaction"):
        model(batch, training=True)


# ------------------------------------------------------------------------------------------------------------------
# Fixture replay THROUGH the drop-in (VERDICT r5 next #1d).  The fixtures under tests/golden/ are what the unmodified
# reference's Model.forward (torch/model/base.py:544-598 -> Head.forward :371-425 -> SequentialBlock.forward,
# block/base.py:236-262) returned on the CPU -- loss, predictions, labels, every parameter gradient -- together with the
# draws it took (oracle/make_golden.py).  Here a model with the reference's module tree and state_dict names gets its three
# hot-path modules class-swapped by dropin.convert_model, loads the fixture's state_dict, replays the draws and must return
# what the reference returned.  Unlike tests/test_dropin_reference_gpu.py this needs no reference source tree (which, being
# Python source, may not travel to a GPU box), so it runs in the driver's round-end GPU suite: the head of THIS round's tree
# behind the drop-in against reference outputs.
FIXTURE_CASES = {
    "xlnet_mlm_item_train": dict(emb_default=32),
    "xlnet_mlm_multi_train": dict(cats=(("category", 40), ("brand", 9)), conts=("price", "age"), d_output=32,
                                  embedding_dims={"item_id": 16, "category": 24, "brand": 8}),
    "xlnet_clm_item_train": dict(masking="clm", emb_default=32, weight_tying=False),
    "gpt2_clm_item_train": dict(masking="clm", emb_default=32, arch="gpt2"),
    "bert_mlm_item_train": dict(emb_default=32, arch="bert"),
}


def _converted_fixture_model(name, params_from=None):
    import golden_utils as gu
    import test_e2e_gpu as e2e
    import transformers4rec_amd as tr
    from transformers4rec_amd import dropin

    d = gu.load(name, params_from)
    model = e2e.build_model(d, **FIXTURE_CASES[params_from or name])
    e2e.load_reference_state(model, d)
    model.to("cuda")
    ns = types.SimpleNamespace(TabularSequenceFeatures=tr.TabularSequenceFeatures, TransformerBlock=tr.TransformerBlock,
                               NextItemPredictionTask=tr.NextItemPredictionTask)
    keys = list(model.state_dict().keys())
    dropin.convert_model(model, ns)
    assert all(getattr(m, "_t4r_hip", False) for m in (model.input_features, model.transformer_block, model.prediction_task))
    assert list(model.state_dict().keys()) == keys
    return gu, d, model


@pytest.mark.parametrize("name", sorted(FIXTURE_CASES))
def test_reference_fixture_replayed_through_the_dropin_train(name):
    gu, d, model = _converted_fixture_model(name)
    model.train()
    feats = model.input_features
    if "draw/bern" in d:
        feats.hip_shadow().masking.set_draws(gu.t(d["draw/bern"]).cuda().to(torch.uint8), gu.t(d["draw/j1"]).cuda(),
                                             gu.t(d["draw/j2"]).cuda())
    x = {k[3:]: gu.t(v).cuda() for k, v in d.items() if k.startswith("in/")}
    out = model(x, training=True)
    assert torch.equal(feats.masking.mask_schema.cpu(), gu.t(d["out/mask_schema"]))
    assert torch.equal(feats.masking.masked_targets.cpu(), gu.t(d["out/masked_targets"]))
    assert torch.equal(out["labels"].cpu(), gu.t(d["out/labels"]))
    pred, loss = out["predictions"].detach().cpu(), float(out["loss"])
    assert float((pred - gu.t(d["out/predictions"])).abs().max()) < 1e-3 and abs(loss - float(d["out/loss"])) < 1e-3   # north_star
    torch.testing.assert_close(pred, gu.t(d["out/predictions"]), rtol=1e-4, atol=5e-5)
    assert abs(loss - float(d["out/loss"])) < 5e-5
    out["loss"].backward()
    g = gu.section(d, "g/")
    named = dict(model.named_parameters())
    for k, ref in g.items():
        assert named[k].grad is not None, f"no gradient reached {k}"
        torch.testing.assert_close(named[k].grad.cpu(), ref, rtol=2e-4, atol=1e-4, msg=lambda m, k=k: f"{k}: {m}")
    assert len(g) > 10
    print(f"[fixture-through-dropin] {name}: loss {loss:.6f} (reference {float(d['out/loss']):.6f}), "
          f"max |d predictions| {float((pred - gu.t(d['out/predictions'])).abs().max()):.2e}, {len(g)} gradients")


@pytest.mark.parametrize("name,train", [("xlnet_mlm_item_eval", "xlnet_mlm_item_train"), ("xlnet_mlm_item_infer", "xlnet_mlm_item_train"),
                                        ("xlnet_clm_item_eval", "xlnet_clm_item_train"), ("xlnet_clm_item_infer", "xlnet_clm_item_train")])
def test_reference_fixture_replayed_through_the_dropin_eval_and_inference(name, train):
    gu, d, model = _converted_fixture_model(name, train)
    model.eval()
    x = {k[3:]: gu.t(v).cuda() for k, v in d.items() if k.startswith("in/")}
    with torch.no_grad():
        out = model(x, testing=True) if name.endswith("_eval") else model(x)
    if name.endswith("_eval"):
        assert torch.equal(out["labels"].cpu(), gu.t(d["out/labels"]))
        torch.testing.assert_close(out["predictions"].cpu(), gu.t(d["out/predictions"]), rtol=1e-4, atol=5e-5)
        assert abs(float(out["loss"]) - float(d["out/loss"])) < 5e-5
    else:
        torch.testing.assert_close(out.cpu(), gu.t(d["out/predictions"]), rtol=1e-4, atol=5e-5)
