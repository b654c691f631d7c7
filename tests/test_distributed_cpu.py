"""N>1 path on CPU: world_size-2 `gloo` run of the data-parallel gradient exchange
(transformers4rec_amd.distributed.GradReducer over transformers4rec_amd.optim.FlatParams buffers).
Semantics under test: per-rank loss = mean over the rank's OWN label rows, gradients
summed over ranks then scaled by 1/world_size -- exactly what torch DDP does for the reference.
The per-rank gradients come from the CPU oracle here (the HIP kernels need a GPU); the reducer,
bucket layout and sharding code are the product code."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import t4r_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tiny_params(seed):
    g = torch.Generator().manual_seed(seed)
    D, n, dh, V = 16, 2, 8, 50
    r = lambda *s: (0.1 * torch.randn(*s, generator=g)).requires_grad_()
    layer = dict(q=r(D, n, dh), k=r(D, n, dh), v=r(D, n, dh), o=r(D, n, dh), r=r(D, n, dh), r_w_bias=r(n, dh),
                 r_r_bias=r(n, dh), ln_w=(1 + 0.1 * torch.randn(D, generator=g)).requires_grad_(), ln_b=r(D),
                 w1=r(4 * D, D), b1=r(4 * D), w2=r(D, 4 * D), b2=r(D),
                 ff_ln_w=(1 + 0.1 * torch.randn(D, generator=g)).requires_grad_(), ff_ln_b=r(D))
    return dict(tables={"item_id": r(V, D)}, masked_item_embedding=r(D), layers=[layer], soft={}, proj=None,
                task_proj=None, output_layer=None), V


def _local_grads(params, ids):
    mask, labels = O.mlm_targets_eval(ids, eval_on_last_item_seq_only=False)  # deterministic labels
    out = O.session_forward(params, dict(n_head=2, eps=0.03, item="item_id", masking="mlm"), {"item_id": ids},
                            mask, labels, True, False)
    out["loss"].backward()
    return out["loss"].detach()


def _named(params):
    named = [("tables.item_id", params["tables"]["item_id"]), ("memb", params["masked_item_embedding"])]
    named += [(f"layer0.{k}", v) for k, v in params["layers"][0].items()]
    return named


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transformers4rec_amd.distributed import GradReducer, shard_batch
    from transformers4rec_amd.optim import FlatParams

    torch.manual_seed(0)
    params, V = _tiny_params(1)                      # identical replicas
    g = torch.Generator().manual_seed(7)
    B, L = 8, 6
    lens = torch.randint(2, L + 1, (B,), generator=g)
    ids_all = torch.randint(1, V, (B, L), generator=g) * (torch.arange(L)[None] < lens[:, None])
    lo, hi = shard_batch(B, rank, world)
    named = _named(params)
    tables = FlatParams([x for x in named if x[0].startswith("tables")])
    dense = FlatParams([x for x in named if not x[0].startswith("tables")])
    loss = _local_grads(params, ids_all[lo:hi])      # grads land in the flat buffers (views)
    assert float(dense.grad.abs().sum()) > 0 and float(tables.grad.abs().sum()) > 0
    red = GradReducer(dense.grad, tables.grad)
    red.reduce_tables_async()
    red.reduce_all(tables_already_launched=True)
    dense.grad.mul_(red.grad_scale)
    tables.grad.mul_(red.grad_scale)
    if rank == 0:
        ret["dense"], ret["tables"], ret["loss0"] = dense.grad.clone(), tables.grad.clone(), float(loss)
        ret["names"] = [n for n, _, _ in dense.entries]
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_gloo_matches_average_of_rank_gradients():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    # single-process expectation: average of the two ranks' gradients (NOT the gradient of the
    # global-mean loss: label counts differ per rank)
    from transformers4rec_amd.distributed import shard_batch
    from transformers4rec_amd.optim import FlatParams

    g = torch.Generator().manual_seed(7)
    B, L = 8, 6
    exp_dense, exp_tables = None, None
    lens = torch.randint(2, L + 1, (B,), generator=g)
    V = 50
    ids_all = torch.randint(1, V, (B, L), generator=g) * (torch.arange(L)[None] < lens[:, None])
    for rank in range(world):
        params, _ = _tiny_params(1)
        named = _named(params)
        tables = FlatParams([x for x in named if x[0].startswith("tables")])
        dense = FlatParams([x for x in named if not x[0].startswith("tables")])
        lo, hi = shard_batch(B, rank, world)
        _local_grads(params, ids_all[lo:hi])
        exp_dense = dense.grad.clone() if exp_dense is None else exp_dense + dense.grad
        exp_tables = tables.grad.clone() if exp_tables is None else exp_tables + tables.grad
    torch.testing.assert_close(ret["dense"], exp_dense / world, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(ret["tables"], exp_tables / world, rtol=1e-6, atol=1e-7)
    assert ret["names"][0] == "memb"


def test_shard_batch_contract():
    from transformers4rec_amd.distributed import shard_batch

    assert [shard_batch(8192, r, 8) for r in (0, 7)] == [(0, 1024), (7168, 8192)]
    with pytest.raises(ValueError):
        shard_batch(10, 0, 4)


def test_flat_params_views_and_regrad():
    from transformers4rec_amd.optim import FlatParams

    a, b = torch.nn.Parameter(torch.randn(3, 5)), torch.nn.Parameter(torch.randn(7))
    a0 = a.detach().clone()
    f = FlatParams([("a", a), ("b", b)])
    assert torch.equal(a.detach(), a0) and a.data_ptr() == f.data.data_ptr()
    assert b.data_ptr() == f.data.data_ptr() + 4 * 16      # 15 floats padded to 16
    (a.sum() * 2 + b.sum()).backward()
    assert float(f.grad[:15].sum()) == 30.0 and float(f.grad[16:23].sum()) == 7.0
    a.grad = None
    f.ensure_grads()
    assert a.grad.data_ptr() == f.grad.data_ptr()


# ------------------------------------------------------------------------------------------ row-sparse exchange
def _cpu_apply(d_table, ids, rows, padding_idx):
    """what ops.scatter_rows_sorted does on the GPU: ids equal to padding_idx or out of range carry nothing"""
    keep = (ids != padding_idx) & (ids >= 0) & (ids < d_table.shape[0])
    d_table.index_add_(0, ids[keep], rows[keep])


def _sparse_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transformers4rec_amd.distributed import GradReducer, SparseRowExchange

    V, D = 40, 6
    g = torch.Generator().manual_seed(100 + rank)
    table = torch.nn.Parameter(torch.zeros(V, D))
    table.grad = torch.randn(V, D, generator=g)                 # the dense part (a tied head's d W)
    dense = torch.randn(11, generator=g)
    dense_part = table.grad.clone()
    # two sparse contributions with DIFFERENT sizes per rank: the lookup scatter (ids with padding 0,
    # rows inside a wider concatenated gradient) and a sampled head's rows (no padding id)
    n1 = 9 + rank
    ids1 = torch.randint(0, V, (n1,), generator=g)
    wide = torch.randn(n1, 10, generator=g)
    n2 = 5 + 2 * rank
    ids2 = torch.randint(0, V, (n2,), generator=g)
    rows2 = torch.randn(n2, D, generator=g)
    sparse = SparseRowExchange(apply_fn=_cpu_apply)
    sparse.attach(table)
    assert table._t4r_sparse_sink is sparse

    class _Ops:      # the column slice the HIP op would take
        pass
    rows1 = wide[:, 2: 2 + D].contiguous()
    sparse.add_rows(table, ids1, rows1, padding_idx=0)
    sparse.add_rows(table, ids2, rows2, padding_idx=-1)
    red = GradReducer(dense, table.grad, sparse=sparse)
    red.reduce_tables_async()
    red.reduce_all()
    # expectation, computed the dense way: every rank scatters locally, then one dense all-reduce
    local = dense_part.clone()
    _cpu_apply(local, ids1, rows1, 0)
    _cpu_apply(local, ids2, rows2, -1)
    dist.all_reduce(local, op=dist.ReduceOp.SUM)
    gathered = [torch.zeros_like(table.grad) for _ in range(world)]
    dist.all_gather(gathered, table.grad)
    if rank == 0:
        ret["got"], ret["want"] = table.grad.clone(), local
        ret["identical_on_all_ranks"] = all(torch.equal(gathered[0], t) for t in gathered)
        ret["bytes"] = sparse.bytes_exchanged
    dist.barrier()
    dist.destroy_process_group()


def test_row_sparse_exchange_equals_dense_reduce():
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_sparse_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    torch.testing.assert_close(ret["got"], ret["want"], rtol=1e-6, atol=1e-6)
    assert ret["identical_on_all_ranks"], "replicas would drift: the sparse apply must give the same bits everywhere"
    assert ret["bytes"] > 0


def _equal_sizes_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transformers4rec_amd.distributed import GradReducer, SparseRowExchange

    V, D, n = 30, 4, 12
    g = torch.Generator().manual_seed(7 + rank)
    table = torch.nn.Parameter(torch.zeros(V, D))
    table.grad = torch.zeros(V, D)
    ids, rows = torch.randint(0, V, (n,), generator=g), torch.randn(n, D, generator=g)
    sp = SparseRowExchange(apply_fn=_cpu_apply, equal_sizes=True).attach(table)
    sp.add_rows(table, ids, rows, padding_idx=0)
    GradReducer(torch.zeros(3), table.grad, sparse=sp).reduce_all()
    ret[rank] = dict(grad=table.grad.clone(), ids=ids, rows=rows)
    dist.destroy_process_group()


def test_sparse_exchange_equal_sizes_skips_the_size_gather():
    """equal_sizes=True (every rank brings the same number of rows): same result as the dense sum, no size exchange"""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_equal_sizes_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    want = torch.zeros(30, 4)
    for r in range(world):
        keep = ret[r]["ids"] != 0
        want.index_add_(0, ret[r]["ids"][keep], ret[r]["rows"][keep])
    for r in range(world):
        assert torch.allclose(ret[r]["grad"], want, atol=1e-6)
    assert torch.equal(ret[0]["grad"], ret[1]["grad"])


def test_sparse_exchange_single_process_is_local_apply():
    from transformers4rec_amd.distributed import GradReducer, SparseRowExchange

    table = torch.nn.Parameter(torch.zeros(7, 3))
    sp = SparseRowExchange(apply_fn=_cpu_apply).attach(table)
    sp.add_rows(table, torch.tensor([1, 0, 1, 6]), torch.ones(4, 3), padding_idx=0)
    GradReducer(torch.zeros(2), None, sparse=sp).reduce_all()
    assert table.grad[1].tolist() == [2.0] * 3 and table.grad[0].tolist() == [0.0] * 3 and table.grad[6].tolist() == [1.0] * 3
    SparseRowExchange.detach(table)
    assert not hasattr(table, "_t4r_sparse_sink")


# ------------------------------------------------------------------------------------------ bench.py's N > 1 wiring
class _StubTask(torch.nn.Module):
    def resolve_head_mode(self, n, v):
        return "materialize"


class _StubModel(torch.nn.Module):
    """stands in for the HIP model: a tied 'table' (dense head gradient + row-sparse lookup scatter through
    the sink, like features._table_scatter) and a dense block whose output carries the head-backward hook"""

    def __init__(self, V=30, D=4):
        super().__init__()
        self.table = torch.nn.Parameter(torch.randn(V, D))
        self.transformer_block = torch.nn.Linear(D, D)
        self.seen_seeds = None

    def forward(self, x, training=True):
        ids = x["item_id"]
        emb = self.table.detach()[ids].requires_grad_()            # lookups are NOT tracked by autograd, as on HIP
        h = self.transformer_block(emb)
        logits = h.reshape(-1, h.shape[-1]) @ self.table.t()        # tied head: dense d table through autograd
        labels = ids.reshape(-1)
        loss = torch.nn.functional.cross_entropy(logits, labels)

        def scatter(g):                                             # what features._table_scatter does
            sink = getattr(self.table, "_t4r_sparse_sink", None)
            rows = g.reshape(-1, g.shape[-1])
            if sink is not None:
                sink.add_rows(self.table, labels, rows, padding_idx=0)
            else:
                self.table.grad.index_add_(0, labels, rows)
        emb.register_hook(scatter)
        return {"loss": loss, "labels": labels}


class _StubOpt:
    def __init__(self):
        self.scales = []

    def step(self, grad_scale=1.0):
        self.scales.append(grad_scale)


def _bench_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import transformers4rec_amd as tr
    from transformers4rec_amd.optim import FlatParams

    torch.manual_seed(0)
    model = _StubModel()
    tables = FlatParams([("table", model.table)])
    dense = FlatParams([(n, p) for n, p in model.transformer_block.named_parameters()])
    sparse = tr.SparseRowExchange(apply_fn=_cpu_apply).attach(model.table)
    reducer = tr.GradReducer(dense.grad, tables.grad, sparse=sparse)
    launched = []
    orig = reducer.reduce_tables_async
    reducer.reduce_tables_async = lambda: (launched.append(float(tables.grad.abs().sum())), orig())[1]
    tr.head_backward_hook(model, reducer.reduce_tables_async)
    g = torch.Generator().manual_seed(50 + rank)                   # every rank its own data
    batches = [{"item_id": torch.randint(1, 30, (4, 5), generator=g)} for _ in range(3)]
    opt = _StubOpt()
    step = bench.make_train_step(model, batches, reducer, opt)
    dense.grad.zero_(); tables.grad.zero_()
    out = step(0)
    # after one step every rank holds the SAME (summed) gradients
    both = [torch.zeros_like(tables.grad) for _ in range(world)]
    dist.all_gather(both, tables.grad)
    same_tables = torch.equal(both[0], both[1])
    dboth = [torch.zeros_like(dense.grad) for _ in range(world)]
    dist.all_gather(dboth, dense.grad)
    dt, out, n_lab = bench.timed_region(step, 1, 3, world, "cpu", first_step=1)
    if rank == 0:
        ret.update(same_tables=same_tables, same_dense=torch.equal(dboth[0], dboth[1]), launched=list(launched[:1]),
                   scales=list(opt.scales), dt=dt, n_lab=n_lab, seeds=[bench.rank_seeds(r) for r in range(world)])
    dist.barrier()
    dist.destroy_process_group()


def _bench_dense_worker(rank, world, port, ret):
    """the "dense" table exchange of bench.setup_data_parallel: local scatter, one late table all-reduce"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import transformers4rec_amd as tr
    from transformers4rec_amd.optim import FlatParams

    res = {}
    for mode in ("dense", "sparse"):
        torch.manual_seed(0)
        model = _StubModel()
        tables = FlatParams([("table", model.table)])
        dense = FlatParams([(n, p) for n, p in model.transformer_block.named_parameters()])
        if mode == "dense":
            reducer, hook = bench.setup_data_parallel(tr, model, dense, tables, world, mode="dense")
            assert hook is None and reducer.sparse is None and not hasattr(model.table, "_t4r_sparse_sink")
        else:
            sparse = tr.SparseRowExchange(apply_fn=_cpu_apply).attach(model.table)
            reducer = tr.GradReducer(dense.grad, tables.grad, sparse=sparse)
            tr.head_backward_hook(model, reducer.reduce_tables_async)
        g = torch.Generator().manual_seed(50 + rank)
        batches = [{"item_id": torch.randint(1, 30, (4, 5), generator=g)}]
        step = bench.make_train_step(model, batches, reducer, _StubOpt())
        dense.grad.zero_(); tables.grad.zero_()
        step(0)
        res[mode] = (tables.grad.clone(), dense.grad.clone())
        tr.SparseRowExchange.detach(model.table)
    if rank == 0:
        ret["dt"] = float((res["dense"][0] - res["sparse"][0]).abs().max())
        ret["dd"] = float((res["dense"][1] - res["sparse"][1]).abs().max())
        ret["norm"] = float(res["dense"][0].abs().max())
    dist.barrier()
    dist.destroy_process_group()


def test_dense_and_sparse_table_exchange_agree():
    """the two ways of moving the table gradient between ranks are switchable and equivalent"""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_bench_dense_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret["norm"] > 0 and ret["dt"] < 1e-6 and ret["dd"] < 1e-6


def test_bench_train_step_wiring_world2():
    """bench.py's own make_train_step / timed_region / head-backward hook / rank seeds on 2 gloo ranks"""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_bench_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret["same_tables"] and ret["same_dense"]
    assert ret["launched"] and ret["launched"][0] > 0      # the table all-reduce started once the head's dense d W existed
    assert ret["scales"] == [0.5] * 5                        # 1/world folded into the optimizer, every step
    assert ret["dt"] > 0 and ret["n_lab"] == 3 * 20
    assert len({s[0] for s in ret["seeds"]}) == world and len({s[1] for s in ret["seeds"]}) == world


# ------------------------------------------------------------------------------------------ cross-rank evaluation metrics
def _metrics_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import transformers4rec_amd as tr

    g = torch.Generator().manual_seed(5)
    ranks_all = torch.randint(0, 40, (101,), generator=g)          # target ranks of 101 label rows
    cut = [0, 37, 101]                                              # uneven shards (eval loaders keep every row)
    task = tr.NextItemPredictionTask(weight_tying=True)
    mine = ranks_all[cut[rank]: cut[rank + 1]]
    if rank == 0:                                                   # two batches on one rank, one on the other
        task.metrics_from_ranks(mine[:10])
        task.metrics_from_ranks(mine[10:])
    else:
        task.metrics_from_ranks(mine)
    agg = task.compute_metrics()
    # a rank that evaluated nothing still takes part in the collective and contributes zeros
    idle = tr.NextItemPredictionTask(weight_tying=True)
    idle.pre = type("P", (), {"module": type("M", (), {"output_weights": torch.zeros(1)})()})()
    if rank == 0:
        idle.metrics_from_ranks(ranks_all)
    agg_idle = idle.compute_metrics()
    ret[rank] = (agg, agg_idle)
    dist.barrier()
    dist.destroy_process_group()


def test_compute_metrics_reduces_over_ranks():
    """Recall / NDCG / AvgPrecision at N > 1 = the mean over EVERY rank's label rows (the reference cat-syncs the
    torchmetrics state, ranking_metric.py:50; trainer.py:519-525), identical on all ranks"""
    import transformers4rec_amd as tr

    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_metrics_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    g = torch.Generator().manual_seed(5)
    ranks_all = torch.randint(0, 40, (101,), generator=g)
    single = tr.NextItemPredictionTask(weight_tying=True)
    single.metrics_from_ranks(ranks_all)
    want = single.compute_metrics()
    assert set(want) == {f"next-item/{m}_{k}" for m in ("ndcg_at", "avg_precision_at", "recall_at") for k in (10, 20)}
    for r in range(world):
        agg, agg_idle = ret[r]
        for k, v in want.items():
            assert abs(agg[k] - v) < 1e-12 and abs(agg_idle[k] - v) < 1e-12, (r, k)
    # closed forms on the full set
    hit20 = (ranks_all < 20).double()
    assert abs(want["next-item/recall_at_20"] - float(hit20.mean())) < 1e-12
    assert abs(want["next-item/avg_precision_at_20"] - float((hit20 / (ranks_all.double() + 1)).mean())) < 1e-6


class _GuardedLinear(torch.nn.Linear):
    """a drop-in-shaped module for the CPU: carries the drop-in marker and runs the drop-in forward's guard
    (dropin._HipFeaturesMixin.forward starts with exactly this call) before its own arithmetic"""
    _t4r_hip = True

    def forward(self, x, training=False):
        from transformers4rec_amd import dropin

        dropin._check_data_parallel(self, training)
        return super().forward(x)


def _sync_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transformers4rec_amd import dropin

    torch.manual_seed(0)
    model = torch.nn.Sequential(_GuardedLinear(4, 3), _GuardedLinear(3, 2))
    model.train()
    x = torch.full((5, 4), float(rank + 1))
    # the guard: a training forward of a drop-in module on > 1 ranks raises until somebody owns the gradient exchange
    dropin.allow_data_parallel(False)
    try:
        model(x)
        guarded = False
    except RuntimeError as e:
        guarded = "enable_data_parallel" in str(e) and "sync_gradients" in str(e)
    # the documented recipe: acknowledge BEFORE the first forward, then forward -> backward -> sync -> step
    dropin.enable_data_parallel(model)
    model(x).sum().backward()
    model[1].bias.grad = None                                       # a parameter without gradient on this rank
    local = [None if p.grad is None else p.grad.clone() for p in model.parameters()]
    dropin.sync_gradients(model)
    synced = [p.grad.clone() for p in model.parameters()]
    model(x)                                                        # still acknowledged: per model, not a side effect of sync
    # ... and sync_gradients itself acknowledges nothing: another model of this process is still guarded
    other = torch.nn.Sequential(_GuardedLinear(4, 2)).train()
    try:
        other(x)
        sticky = True
    except RuntimeError:
        sticky = False
    ret[rank] = (guarded and not sticky, local, synced)
    dist.barrier()
    dist.destroy_process_group()


def test_dropin_sync_gradients_and_ddp_guard():
    """the HIP backward bypasses autograd hooks, so DDP would never reduce the drop-in's gradients:
    the forward raises on world_size > 1 until the caller has acknowledged (per model, BEFORE the first forward:
    `enable_data_parallel` / `convert_model(data_parallel=True)`) that it runs `sync_gradients` (flat averaged all-reduce
    of every .grad) between backward and step -- the recipe of INTEGRATION.md, run end to end on two gloo ranks"""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_sync_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret[0][0] and ret[1][0]
    for i in range(4):
        l0, l1 = ret[0][1][i], ret[1][1][i]
        want = (torch.zeros_like(ret[0][2][i]) if l0 is None else l0) * 0.5 + (torch.zeros_like(ret[0][2][i]) if l1 is None else l1) * 0.5
        torch.testing.assert_close(ret[0][2][i], want)
        assert torch.equal(ret[0][2][i], ret[1][2][i])
