"""The drop-in boundary, executed: transformers4rec_amd.dropin's
subclasses of the REFERENCE's TabularSequenceFeatures / TransformerBlock / NextItemPredictionTask
are built by the reference's own constructors (`from_schema`, `XLNetConfig.to_torch_model`, through the
reference's Model / Head / SequentialBlock), pass every isinstance gate, leave the state_dict untouched
and share (not copy) every parameter with the HIP-side shadow.

Needs the reference source tree (imported through oracle/ref_standins.py): runs in the build
container, skipped where /root/reference does not exist (the GPU box).  The GPU half of the adapter
(forward / backward through the shadow) is tests/test_dropin_gpu.py.
"""
import os

import pytest
import torch

import ref_standins as rs

pytestmark = pytest.mark.skipif(not os.path.isdir(rs.REFERENCE_ROOT), reason="reference source tree not present")


@pytest.fixture(scope="module")
def tr():
    return rs.import_reference()


@pytest.fixture()
def hip(tr):
    from transformers4rec_amd import dropin

    classes = dropin.install(tr)
    yield classes
    dropin.uninstall(tr)


def _schema(V, L, cats=(), conts=()):
    import make_golden as mg

    return mg.make_schema(V, L, cats, conts)


def _build(tr, arch="xlnet", masking="mlm", cats=(), conts=(), d_output=None, weight_tying=True, sampled=False,
           emb=32, aggregation="concat", **fkw):
    V, L, d = 300, 20, 32
    schema = _schema(V, L, cats, conts)
    kw = dict(max_sequence_length=L, masking=masking, aggregation=aggregation, embedding_dim_default=emb)
    if conts:
        kw["continuous_soft_embeddings"] = True
    if d_output:
        kw["d_output"] = d_output
    kw.update(fkw)
    torch.manual_seed(0)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, **kw)
    from transformers4rec.config import transformer as tconf

    cfgc = {"xlnet": tconf.XLNetConfig, "gpt2": tconf.GPT2Config, "bert": tconf.BertConfig}[arch]
    cfg = cfgc.build(d_model=d, n_head=2, n_layer=2, total_seq_length=L)
    task = tr.NextItemPredictionTask(weight_tying=weight_tying, sampled_softmax=sampled, max_n_samples=20)
    return cfg.to_torch_model(inputs, task), inputs, task


CASES = {
    "xlnet_mlm_tied": dict(),
    "xlnet_mlm_multi_proj": dict(cats=(("category", 40), ("brand", 9)), conts=("price", "age"), d_output=32,
                                 embedding_dims={"item_id": 16, "category": 24, "brand": 8}),
    "xlnet_clm_untied": dict(masking="clm", weight_tying=False),
    "xlnet_mlm_sum_sampled": dict(cats=(("category", 40),), aggregation="element-wise-sum", sampled=True),
    "gpt2_clm": dict(arch="gpt2", masking="clm"),
    "bert_mlm": dict(arch="bert"),
    "xlnet_mlm_task_block": dict(d_output=32, embedding_dims={"item_id": 24}),   # hidden 32 != item dim 24
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_builders_produce_hip_modules(tr, hip, name):
    HipF, HipB, HipT = hip
    model, inputs, task = _build(tr, **CASES[name])
    body = model.heads[0].body
    # the gates of the reference: block/base.py:149-154,182 ; config/transformer.py:113 ; model/base.py:410
    assert type(inputs) is HipF and isinstance(inputs, tr.TabularSequenceFeatures)
    assert type(body[1]) is HipB and isinstance(body[1], tr.TransformerBlock)
    assert type(task) is HipT and isinstance(task, tr.NextItemPredictionTask) and isinstance(task, tr.PredictionTask)
    assert isinstance(model, tr.Model) and isinstance(model.heads[0], tr.Head)
    assert isinstance(body, tr.SequentialBlock) and body.inputs is inputs
    # the task was wired by the reference's Head.build: same masking object in all three modules
    assert task.masking is inputs.masking and body[1].masking is inputs.masking
    keys_before = list(model.state_dict().keys())

    fs, bs, ts = inputs.hip_shadow(), body[1].hip_shadow(), task.hip_shadow()
    assert list(model.state_dict().keys()) == keys_before, "the shadows must not register anything"
    assert inputs.hip_shadow() is fs
    # every shadow parameter IS a reference parameter (same object), nothing left on the meta device
    ref_ids = {id(p) for p in model.parameters()}
    for label, sh in (("features", fs), ("block", bs), ("task", ts)):
        ps = list(sh.parameters())
        assert ps, label
        for n, p in sh.named_parameters():
            assert id(p) in ref_ids, f"{label}: {n} is not shared with the reference model"
            assert p.device.type != "meta", f"{label}: {n} left on meta"
        for n, b in sh.named_buffers():
            assert b.device.type != "meta", f"{label}: buffer {n} left on meta"
    # shared names resolve to the same tensors
    assert fs.item_embedding_table.weight is inputs.item_embedding_table.weight
    assert fs.masking.masked_item_embedding is inputs.masking.masked_item_embedding
    W_ref = task.pre.module.item_embedding_table.weight if task.weight_tying else task.pre.module.output_layer
    assert ts.pre.module.output_weights is W_ref
    if task.task_block is not None:
        assert ts.task_block[0][0].weight is task.task_block[0][0].weight
    # configuration carried over
    assert fs._aggregation == CASES[name].get("aggregation", "concat")
    assert type(fs.masking).__name__ == type(inputs.masking).__name__
    assert bs.transformer.config.hidden_size == body[1].transformer.config.hidden_size


def test_state_dict_round_trip_both_ways(tr, hip):
    from transformers4rec_amd import dropin

    model, inputs, task = _build(tr, **CASES["xlnet_mlm_multi_proj"])
    dropin.uninstall(tr)
    plain, _, _ = _build(tr, **CASES["xlnet_mlm_multi_proj"])      # the unmodified reference classes
    assert type(plain.heads[0].body[0]).__name__ == "TabularSequenceFeatures"
    sd_hip, sd_ref = model.state_dict(), plain.state_dict()
    assert list(sd_hip.keys()) == list(sd_ref.keys())
    assert all(sd_hip[k].shape == sd_ref[k].shape for k in sd_ref)
    with torch.no_grad():
        for p in plain.parameters():
            p.add_(1.0)
    inputs.hip_shadow()                                   # built BEFORE the load: loading must reach it
    model.load_state_dict(plain.state_dict())             # reference -> hip
    assert torch.equal(inputs.hip_shadow().item_embedding_table.weight, plain.heads[0].body[0].item_embedding_table.weight)
    plain2, _, _ = _build(tr, **CASES["xlnet_mlm_multi_proj"])
    plain2.load_state_dict(model.state_dict())            # hip -> reference
    for (k, a), (_, b) in zip(plain2.state_dict().items(), plain.state_dict().items()):
        assert torch.equal(a, b), k


def test_convert_model_in_place(tr):
    from transformers4rec_amd import dropin

    model, inputs, task = _build(tr)
    assert not getattr(inputs, "_t4r_hip", False)
    keys = list(model.state_dict().keys())
    ids = [id(p) for p in model.parameters()]
    dropin.convert_model(model, tr)
    body = model.heads[0].body
    assert all(getattr(m, "_t4r_hip", False) for m in (body[0], body[1], task))
    assert isinstance(body[0], tr.TabularSequenceFeatures) and isinstance(task, tr.NextItemPredictionTask)
    assert list(model.state_dict().keys()) == keys and [id(p) for p in model.parameters()] == ids
    assert task.hip_shadow().pre.module.output_weights is inputs.item_embedding_table.weight


def test_no_cpu_fallback(tr, hip):
    """a CPU batch must fail loudly in the HIP modules, not run the reference's torch code"""
    from transformers4rec_amd import _lib

    model, inputs, task = _build(tr)
    x = {"item_id": torch.randint(1, 300, (4, 20))}
    with pytest.raises(_lib.T4RHipError):
        inputs(x, training=True)
    with pytest.raises(_lib.T4RHipError):
        model.heads[0].body[1](torch.zeros(4, 20, 32))


def test_off_path_configurations_raise(tr, hip):
    schema = _schema(300, 20)
    torch.manual_seed(0)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=20, masking="plm",
                                                    embedding_dim_default=32)
    with pytest.raises(NotImplementedError, match="masking"):
        inputs.hip_shadow()


def test_shadow_task_keeps_the_users_metric_configuration(tr):
    """the shadow must not replace the reference task's metric list by the default set -- rank-based
    metrics keep their cut-offs, `metrics=[]` stays empty, non-rank metrics are left to the reference"""
    from transformers4rec_amd import dropin
    from transformers4rec.torch import ranking_metric as rm

    model, inputs, task = _build(tr)
    dropin.convert_model(model, tr)
    task.metrics = torch.nn.ModuleList([rm.RecallAt(top_ks=[5], labels_onehot=True)])
    dropin.drop_shadow(task)
    sh = task.hip_shadow()
    assert [(m.name, tuple(m.top_ks)) for m in sh.metrics] == [("recall_at", (5,))]

    class NotRankBased(torch.nn.Module):
        pass

    task.metrics = torch.nn.ModuleList([NotRankBased(), rm.NDCGAt(top_ks=[3, 7], labels_onehot=True)])
    dropin.drop_shadow(task)
    assert [(m.name, tuple(m.top_ks)) for m in task.hip_shadow().metrics] == [("ndcg_at", (3, 7))]
    task.metrics = torch.nn.ModuleList([])
    dropin.drop_shadow(task)
    assert tuple(task.hip_shadow().metrics) == ()


def test_convert_model_data_parallel_acknowledges_per_model(tr):
    from transformers4rec_amd import dropin

    model, inputs, task = _build(tr)
    dropin.convert_model(model, tr, data_parallel=True)
    body = model.heads[0].body
    assert all(m.__dict__.get(dropin._DP_ATTR) for m in (body[0], body[1], task))
    assert not any(dropin._DP_ATTR in k for k in model.state_dict())
    other, inputs2, task2 = _build(tr)
    dropin.convert_model(other, tr)
    assert not task2.__dict__.get(dropin._DP_ATTR, False)
