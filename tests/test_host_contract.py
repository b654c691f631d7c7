"""Host-side mirror of the reference's module contract, checked without a GPU:
state_dict names / shapes against the reference-generated fixtures, constructor and build errors
with the reference's messages, the pre / post transformation registry strings, and the
no-CPU-fallback rule."""
import numpy as np
import pytest
import torch

import golden_utils as gu


def _build(d, **kw):
    import transformers4rec_amd as tr

    L, V = int(d["meta/L"]), int(d["meta/V"])
    cats, conts = kw.pop("cats", ()), kw.pop("conts", ())
    context = kw.pop("context", ())
    arch = kw.pop("arch", "xlnet")
    weight_tying = kw.pop("weight_tying", True)
    sampled, max_n = kw.pop("sampled", False), kw.pop("max_n", 100)
    schema = tr.session_schema(V - 1, L, cats, conts)
    for name, card in context:
        schema = schema + tr.Schema([tr.ColumnSchema(name, [tr.Tags.CATEGORICAL], tr.IntDomain(0, card))])
    fk = dict(max_sequence_length=L, masking=kw.pop("masking", "mlm"), aggregation=kw.pop("aggregation", "concat"))
    if conts:
        fk["continuous_soft_embeddings"] = True
    fk.update(kw)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, **fk)
    ck = dict(d_model=int(d["meta/d_model"]), n_head=int(d["meta/n_head"]), n_layer=int(d["meta/n_layer"]),
              total_seq_length=L)
    cfg = {"xlnet": tr.XLNetConfig, "gpt2": tr.GPT2Config, "bert": tr.BertConfig}[arch].build(**ck)
    return cfg.to_torch_model(inputs, tr.NextItemPredictionTask(weight_tying=weight_tying, sampled_softmax=sampled,
                                                               max_n_samples=max_n))


CASES = {
    "xlnet_mlm_item_train": dict(embedding_dim_default=32),
    "xlnet_mlm_multi_train": dict(cats=(("category", 40), ("brand", 9)), conts=("price", "age"), d_output=32,
                                  embedding_dims={"item_id": 16, "category": 24, "brand": 8}),
    "xlnet_clm_item_train": dict(masking="clm", embedding_dim_default=32, weight_tying=False),
    "xlnet_mlm_sum_sampled_train": dict(cats=(("category", 40),), aggregation="element-wise-sum",
                                        embedding_dim_default=32, sampled=True, max_n=20),
    "xlnet_mlm_context_train": dict(context=(("country", 17),), d_output=32,
                                    embedding_dims={"item_id": 24, "country": 8}),
    "gpt2_clm_item_train": dict(masking="clm", embedding_dim_default=32, arch="gpt2"),
    "bert_mlm_item_train": dict(embedding_dim_default=32, arch="bert"),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_state_dict_names_and_shapes_match_reference(name):
    """every tensor of the reference's state_dict has a home of the same shape (checkpoints interchange)"""
    d = gu.load(name)
    model = _build(d, **dict(CASES[name]))
    own = model.state_dict()
    ref = gu.section(d, "p/")
    assert ref, "fixture without parameters"
    for k, v in ref.items():
        assert k in own, f"reference key without a home: {k}"
        assert tuple(own[k].shape) == tuple(v.shape), (k, tuple(own[k].shape), tuple(v.shape))
    model.load_state_dict({k: v for k, v in ref.items()}, strict=False)
    for k, v in ref.items():
        assert torch.equal(model.state_dict()[k], v)


def test_prepost_state_dict_names():
    import transformers4rec_amd as tr

    d = gu.load("xlnet_mlm_prepost_concat_train")
    schema = tr.session_schema(int(d["meta/V"]) - 1, int(d["meta/L"]), (("category", 40), ("brand", 9)), ("price",))
    inputs = tr.TabularSequenceFeatures.from_schema(
        schema, max_sequence_length=int(d["meta/L"]), masking="mlm", aggregation="concat", d_output=32,
        continuous_soft_embeddings=True, embedding_dims={"item_id": 16, "category": 24, "brand": 8},
        pre=["stochastic-swap-noise"], post=[tr.TabularDropout(0.25), "layer-norm"])
    keys = set(inputs.state_dict())
    want = {k[len("heads.0.body.0."):] for k in gu.section(d, "p/") if k.startswith("heads.0.body.0.")}
    assert want <= keys, sorted(want - keys)
    assert isinstance(inputs.categorical_module.pre, tr.StochasticSwapNoise)
    assert [type(m).__name__ for m in inputs.categorical_module.post] == ["TabularDropout", "TabularLayerNorm"]
    with pytest.raises(ValueError, match="unsupported post transformation"):
        tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=20, continuous_soft_embeddings=True, post="nope")
    with pytest.raises(ValueError):
        tr.TabularDropout(1.0)


def test_constructor_and_build_errors_follow_the_reference():
    import transformers4rec_amd as tr

    schema = tr.session_schema(100, 20, (("category", 10),))
    # element-wise aggregation needs equal dims (tabular/aggregation.py:140-157)
    with pytest.raises(ValueError, match="same dimension"):
        tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=20, aggregation="element-wise-sum",
                                               embedding_dims={"item_id": 16, "category": 8})
    # masking needs an item id (features/sequence.py:225-226)
    no_item = tr.Schema([tr.ColumnSchema("category", [tr.Tags.CATEGORICAL, tr.Tags.LIST], tr.IntDomain(0, 10),
                                         tr.ValueCount(1, 20))])
    with pytest.raises(ValueError, match="item_id"):
        tr.TabularSequenceFeatures.from_schema(no_item, max_sequence_length=20, masking="mlm")
    inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=20, masking="mlm", d_output=32)
    assert inputs.aggregation == "concat" and tuple(inputs.output_size()) == (-1, 20, 32)
    task = tr.NextItemPredictionTask()
    with pytest.raises(ValueError, match="3-dim"):          # prediction_task.py:371-374
        task.build(body=None, input_size=torch.Size([-1, 32]), inputs=inputs)
    nomask = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=20, aggregation="concat")
    with pytest.raises(ValueError, match="masking schema"):   # prediction_task.py:402-404
        tr.NextItemPredictionTask().build(body=None, input_size=torch.Size([-1, 20, 72]), inputs=nomask)
    with pytest.raises(NotImplementedError):
        tr.NextItemPredictionTask(loss=torch.nn.MSELoss())
    # the reference's masking / architecture rules (block/transformer.py:109-134)
    cfg = tr.GPT2Config.build(32, 2, 1, total_seq_length=20)
    with pytest.raises(ValueError, match="is not supported by"):
        tr.TransformerBlock(cfg, masking=inputs.masking)


def test_cpu_tensors_are_rejected_not_silently_computed():
    """no CPU fallback: the product path refuses host tensors instead of computing on them"""
    import transformers4rec_amd as tr
    from transformers4rec_amd import ops

    with pytest.raises(Exception) as e:
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4))
    assert "cuda" in str(e.value).lower() or "device" in str(e.value).lower() or "hip" in str(e.value).lower()
    schema = tr.session_schema(50, 8)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=8, masking="mlm", embedding_dim_default=16)
    with pytest.raises(Exception):
        inputs({"item_id": torch.randint(1, 50, (2, 8))}, training=True)


def test_recall_ndcg_oracle_against_reference_known_answers():
    """ranking_metric known answers of the reference (tests/unit/torch/test_ranking_metrics.py:48-117 shape:
    one relevant item per row) restated on tiny inputs"""
    import t4r_oracle as O

    scores = torch.tensor([[0.9, 0.1, 0.5, 0.3], [0.2, 0.8, 0.7, 0.1]])
    labels = torch.tensor([2, 3])
    assert O.recall_at_k(scores, labels, 2).tolist() == [1.0, 0.0]
    assert O.recall_at_k(scores, labels, 4).tolist() == [1.0, 1.0]
    nd = O.ndcg_at_k(scores, labels, 4)
    assert abs(float(nd[0]) - 1 / np.log2(3)) < 1e-6 and abs(float(nd[1]) - 1 / np.log2(5)) < 1e-6


def test_missing_library_fails_loudly():
    """no silent fallback: with the shared library absent every op raises T4RHipError"""
    import subprocess
    import sys

    code = ("import transformers4rec_amd as tr\n"
            "from transformers4rec_amd import _lib\n"
            "try:\n"
            "    _lib.call('t4r_gemm_f32')\n"
            "except _lib.T4RHipError as e:\n"
            "    print('RAISED', 'no cpu fallback' in str(e).lower() and 'not found' in str(e).lower())\n")
    import os
    env = dict(os.environ, T4R_HIP_LIB="/nonexistent/libt4r_hip.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert "RAISED True" in out.stdout, out.stdout + out.stderr


def test_logits_leading_dimension_is_256_byte_aligned():
    """[N, V] logits live in a buffer whose rows start on 256-byte boundaries (with 16-byte
    aligned rows every 128-byte store segment of the GEMM epilogue straddled two cache lines)."""
    from transformers4rec_amd import ops
    for V in (1, 63, 64, 65, 1001, 100001):
        ld = ops.pad_ld(V)
        assert ld >= V and ld % 64 == 0 and ld - V < 64


def test_rng_defaults_follow_torch_seed_and_rank_and_can_be_checkpointed(monkeypatch):
    """default Philox keys derive from torch.initial_seed() + rank (not 0), the stream positions are
    checkpointable outside the state_dict (whose names are a contract with the reference)."""
    import transformers4rec_amd as tr

    schema = tr.session_schema(50, 8)

    def make():
        inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=8, masking="mlm",
                                                        embedding_dim_default=16)
        return tr.XLNetConfig.build(16, 2, 1, total_seq_length=8).to_torch_model(
            inputs, tr.NextItemPredictionTask(weight_tying=True))

    torch.manual_seed(11)
    a = make()
    sa = (a.input_features.masking.seed, a.transformer_block.transformer.seed)
    torch.manual_seed(11)
    b = make()
    assert (b.input_features.masking.seed, b.transformer_block.transformer.seed) == sa   # reproducible
    assert sa[0] != sa[1] and 0 not in sa                                                  # distinct streams
    torch.manual_seed(12)
    assert make().input_features.masking.seed != sa[0]                                     # follows manual_seed
    monkeypatch.setenv("RANK", "3")
    torch.manual_seed(11)
    assert make().input_features.masking.seed != sa[0]                                     # and the rank
    monkeypatch.delenv("RANK")
    keys = list(a.state_dict().keys())
    a.input_features.masking._rng_offset = 160
    a.transformer_block.transformer._drop_offset = 7
    st = tr.get_rng_state(a)
    assert list(a.state_dict().keys()) == keys, "RNG state must stay out of the state_dict"
    tr.set_rng_state(b, st)
    assert b.input_features.masking._rng_offset == 160 and b.transformer_block.transformer._drop_offset == 7
    assert b.input_features.masking.seed == a.input_features.masking.seed
    with pytest.raises(KeyError):
        tr.set_rng_state(b, {"nope": {"_seed": 1}})


def test_frozen_parameters_keep_no_grad():
    """a frozen parameter's .grad is not touched by the direct-.grad contract"""
    from transformers4rec_amd.masking import _grad_buf

    p = torch.nn.Parameter(torch.zeros(3, 2), requires_grad=False)
    s = _grad_buf(p)
    assert p.grad is None and s.shape == p.shape and _grad_buf(p) is s
    q = torch.nn.Parameter(torch.zeros(3, 2))
    assert _grad_buf(q) is q.grad


def test_construction_time_limits():
    import transformers4rec_amd as tr

    # (round 6: the attention kernels take any length -- csrc/xlnet_attn_long.hip beyond 64 positions --; what bounds an
    # XLNet sequence is the target kernel's 1023 positions)
    schema = tr.session_schema(100, 1023)
    mlm = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=1023, masking="mlm", embedding_dim_default=16)
    clm = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=1023, masking="clm", embedding_dim_default=16)
    cfg = tr.XLNetConfig.build(16, 2, 1, total_seq_length=1023)
    tr.TransformerBlock(cfg, masking=clm.masking)                      # 1023 positions fit
    with pytest.raises(ValueError, match="at most 1023"):              # MLM inference needs L + 1 = 1024
        tr.TransformerBlock(cfg, masking=mlm.masking)
    tr.TransformerBlock(tr.XLNetConfig.build(16, 2, 1, total_seq_length=100), masking=mlm.masking)     # 64 is no limit any more
    # labels are item ids: a smaller target_dim would index out of range in the head kernels
    small = tr.TabularSequenceFeatures.from_schema(tr.session_schema(100, 20), max_sequence_length=20, masking="mlm",
                                                   embedding_dim_default=16)
    with pytest.raises(ValueError, match="target_dim"):
        tr.NextItemPredictionTask(target_dim=50).build(body=None, input_size=torch.Size([-1, 20, 16]), inputs=small)


def test_rng_state_carries_the_tabular_dropout_key():
    """the process-wide TabularDropout key is part of get / set_rng_state, so a resumed run with another
    torch.initial_seed() replays the same post-dropout masks"""
    import transformers4rec_amd as tr
    from transformers4rec_amd import features, rng

    m = torch.nn.Linear(2, 2)
    features.set_post_seed(None)
    torch.manual_seed(21)
    st = tr.get_rng_state(m)
    key = st[rng._POST_KEY]["_seed"]
    assert key == features.post_seed()
    features.set_post_seed(None)
    torch.manual_seed(22)
    assert features.post_seed() != key                     # another global seed -> another key ...
    tr.set_rng_state(m, st)
    assert features.post_seed() == key                     # ... unless the saved state is restored
    features.set_post_seed(None)


def test_loader_shards_training_equal_eval_complete(tmp_path):
    """equal shards (trailing rows dropped, with a warning) only for training; evaluation loaders keep
    every row, shard sizes differing by at most one"""
    import pyarrow as pa
    import pyarrow.parquet as pq
    import transformers4rec_amd as tr

    n = 11
    col = pa.array([[i + 1] * (1 + i % 3) for i in range(n)], type=pa.list_(pa.int64()))
    path = str(tmp_path / "d.parquet")
    pq.write_table(pa.table({"item_id": col}), path)
    with pytest.warns(UserWarning, match="dropped"):
        train = [tr.ParquetSessionLoader(path, 4, 5, device="cpu", global_size=3, global_rank=r) for r in range(3)]
    assert [len(l.dataset) for l in train] == [3, 3, 3]
    ev = [tr.ParquetSessionLoader(path, 4, 5, device="cpu", global_size=3, global_rank=r, drop_uneven=False)
          for r in range(3)]
    assert [len(l.dataset) for l in ev] == [4, 4, 3]
    spans = [(l._row0, l._row0 + l._rows) for l in ev]
    assert spans == [(0, 4), (4, 8), (8, 11)]              # contiguous, complete, disjoint


def test_schema_from_json_text():
    import transformers4rec_amd as tr
    from transformers4rec_amd.schema import categorical_cardinalities

    js = ('{"feature": [{"name": "item_id/list", "type": "INT", "intDomain": {"name": "item_id/list", "min": "1", "max": "51996", '
          '"isCategorical": true}, "valueCount": {"min": "2", "max": "185"}, "annotation": {"tag": ["item_id", "list", '
          '"categorical", "item"]}}, {"name": "user_age", "type": "FLOAT", "annotation": {"tag": ["continuous"]}}]}')
    s = tr.Schema.from_json(js)
    assert s.column_names == ["item_id/list", "user_age"] and s.item_id_column_name == "item_id/list"
    assert categorical_cardinalities(s) == {"item_id/list": 51997}
    assert s.feature[0].value_count.max == 185 and s.select_by_tag(tr.Tags.CONTINUOUS).column_names == ["user_age"]


def test_ranking_metric_argument_forms():
    import transformers4rec_amd as tr
    from transformers4rec_amd import ranking_metric as RM

    class NDCGAt:                                           # stands for the reference's metric object: recognised by name
        top_ks, labels_onehot = [3, 7], True

    t = tr.NextItemPredictionTask(metrics=[NDCGAt(), "map", RM.RecallAt(top_ks=[5])])
    assert [(m.name, m.top_ks) for m in t.metrics] == [("ndcg_at", [3, 7]), ("avg_precision_at", [10, 20]), ("recall_at", [5])]
    out = t.metrics_from_ranks(torch.tensor([0, 4, 6, 30]))
    assert out["recall_at_5"].tolist() == [1.0, 1.0, 0.0, 0.0]
    assert out["avg_precision_at_10"].tolist() == pytest.approx([1.0, 0.2, 1 / 7, 0.0])
    assert t.compute_metrics()["next-item/recall_at_5"] == 0.5

def test_round3_host_queries_of_the_library():
    """host-only entry points added in round 3 (no device work): workspace offsets of the stack prologue, the product counts
    the bench prices the kernels with, the environment switches behind them"""
    import ctypes
    from transformers4rec_amd import _lib

    lib = _lib.load()
    po, ko = ctypes.c_long(-5), ctypes.c_long(-5)
    assert lib.t4r_xlnet_layer_ws_offsets(8, 20, 128, 4, 1, ctypes.addressof(po), ctypes.addressof(ko)) == 0
    total = lib.t4r_xlnet_layer_ws_floats(8, 20, 128, 4, 1)
    planes = lib.t4r_xlnet_layer_planes_floats(128)
    assert 0 <= ko.value < po.value and po.value + planes <= total          # k_r before the planes, both inside the workspace
    assert po.value % 4 == 0 and ko.value % 4 == 0                          # 16-byte aligned regions
    lib.t4r_xlnet_layer_ws_offsets(8, 20, 256, 4, 1, ctypes.addressof(po), ctypes.addressof(ko))
    assert po.value == -1                                                   # no fused kernels (no planes) at this width
    assert planes >= 25 * 3 * 128 * 128 // 2 + 25 * 2 * 128 * 128 // 2      # bf16 planes + fp16 planes (+ scales)
    assert lib.t4r_xlnet_fused_products() in (3, 6) and lib.t4r_head_split_fwd_products() in (3, 6)
    # deferred join switches are plain host state: callable without a device
    lib.t4r_xlnet_layer_bwd_defer(1)
    lib.t4r_xlnet_layer_bwd_defer(0)
    lib.t4r_xlnet_stack_prepared(0)
    assert lib.t4r_apply_mask_bwd_ws_floats(4, 20, 128) == ((4 * 20 + 31) // 32) * 128


def test_table_config_default_initializer_is_the_references():
    """features/embedding.py:460-464 -- TableConfig(initializer=None) means normal_(0, 0.05), applied to bag
    tables too (table_to_embedding_module always runs it); a non-callable initializer is a ValueError"""
    import pytest
    import transformers4rec_amd as tr
    from transformers4rec_amd import features as F

    torch.manual_seed(0)
    cfg = F.TableConfig(vocabulary_size=4000, dim=64, name="t")
    assert callable(cfg.initializer)
    w = torch.empty(4000, 64)
    cfg.initializer(w)
    assert abs(float(w.std()) - 0.05) < 2e-3 and abs(float(w.mean())) < 1e-3
    bag = F._BagTable(4000, 64, "mean", cfg.initializer)
    assert abs(float(bag.weight.std()) - 0.05) < 2e-3
    assert abs(float(F._BagTable(4000, 64).weight.std()) - 0.05) < 2e-3
    with pytest.raises(ValueError, match="callable"):
        F.TableConfig(vocabulary_size=10, dim=4, initializer=0.05)
    feats = tr.EmbeddingFeatures({"a": F.FeatureConfig(cfg)})
    assert abs(float(feats.embedding_tables["a"].weight.std()) - 0.05) < 2e-3


def test_table_config_default_initializer_pickles():
    """the default initializer is a functools.partial as in the reference (features/embedding.py:460-464); a lambda
    made pickle / torch.save(model) fail"""
    import pickle

    import transformers4rec_amd as tr
    from transformers4rec_amd.features import TableConfig

    tc = pickle.loads(pickle.dumps(TableConfig(10, 4)))
    w = torch.zeros(10, 4)
    tc.initializer(w)
    assert 0.0 < float(w.std()) < 0.2
    schema = tr.session_schema(50, 8)
    inputs = tr.TabularSequenceFeatures.from_schema(schema, max_sequence_length=8, masking="mlm", embedding_dim_default=16)
    pickle.loads(pickle.dumps(inputs))


def test_head_mode_by_size(monkeypatch):
    """the size rule every fallback of the head uses: materialise while the fp32 scores fit T4R_HEAD_AUTO_GB, else the chunked head"""
    from transformers4rec_amd.prediction_task import NextItemPredictionTask

    monkeypatch.setenv("T4R_HEAD_AUTO_GB", "4")
    assert NextItemPredictionTask.size_head_mode(2765, 100_001) == "materialize"
    assert NextItemPredictionTask.size_head_mode(15_360, 10_000_001) == "fused"
    monkeypatch.setenv("T4R_HEAD_AUTO_GB", "0.0001")
    assert NextItemPredictionTask.size_head_mode(2765, 100_001) == "fused"
