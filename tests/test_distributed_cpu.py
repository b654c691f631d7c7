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
