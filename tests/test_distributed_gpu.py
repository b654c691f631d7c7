"""The N > 1 training step on REAL kernels: two processes share the one GPU of the test box and exchange gradients
over gloo (RCCL refuses two ranks on one device; the collectives are the only thing swapped out).  bench.py's own
wiring -- setup_data_parallel (row-sparse sink on the tables, async table all-reduce hooked after the head's
backward), make_train_step, FusedAdam with 1/world folded in -- must (a) leave the replicas bit-identical after
several steps and (b) equal a single process that averages the two ranks' gradients itself."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

V, D, NL, NH, L, B, STEPS = 5000, 64, 2, 4, 20, 64, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(device):
    import bench

    return bench.build(device, 0.0, v_items=V, d_model=D, n_layer=NL, n_head=NH, seq=L, lr=1e-2)


def _batches(tr, schema, rank, device):
    return [tr.random_data_from_schema(schema, B, L, seed=100 * rank + i, device=device) for i in range(STEPS)]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench

    tr, schema, model, dense, tables, opt = _build(dev)
    reducer, hook = bench.setup_data_parallel(tr, model, dense, tables, world)
    assert hook is not None and reducer.sparse is not None
    model.input_features.masking.seed, model.transformer_block.transformer.seed = bench.rank_seeds(rank)
    model.train()
    batches = _batches(tr, schema, rank, dev)
    # step 0 by hand (the same calls make_train_step makes) to look at the EXCHANGED gradients before Adam eats them
    out = model(batches[0], training=True)
    out["loss"].backward()
    reducer.reduce_all()
    grads0 = torch.cat([f.grad for f in opt.flats]).cpu() * reducer.grad_scale
    opt.step(grad_scale=reducer.grad_scale)
    step = bench.make_train_step(model, batches, reducer, opt)
    losses = [float(out["loss"].detach())] + [float(step(i)["loss"].detach()) for i in range(1, STEPS)]
    torch.cuda.synchronize()
    flat = torch.cat([f.data for f in opt.flats]).cpu()
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    if rank == 0:
        ret.update(identical=torch.equal(both[0], both[1]), grads0=grads0, losses=losses,
                   bytes=reducer.sparse.bytes_exchanged)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_step_on_one_gpu():
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert ret["identical"], "replicas drifted: the gradient exchange is not the same on both ranks"
    assert ret["bytes"] > 0
    # single-process expectation for step 0: both ranks' batches, gradients averaged.  (Parameters after Adam are
    # not compared across implementations: m / sqrt(v) turns a last-bit difference of a near-zero gradient into a
    # full-size update; the split-K atomics already differ in the last bits from run to run.)
    import bench

    dev = torch.device("cuda", 0)
    tr, schema, model, dense, tables, opt = _build(dev)
    model.train()
    masking, xl = model.input_features.masking, model.transformer_block.transformer
    for r in range(world):                          # replay rank r's mask stream
        masking.seed, xl.seed = bench.rank_seeds(r)
        masking._rng_offset = 0
        out = model(_batches(tr, schema, r, dev)[0], training=True)
        out["loss"].backward()                      # the two ranks' gradients accumulate in the flat buckets
    torch.cuda.synchronize()
    want = torch.cat([f.grad for f in opt.flats]).cpu() / world
    got = ret["grads0"]
    scale = float(want.abs().max())
    assert scale > 0 and float((got - want).abs().max()) < 1e-4 * scale, (float((got - want).abs().max()), scale)
    assert ret["losses"][-1] < ret["losses"][0] + 0.5
