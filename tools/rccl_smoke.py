#!/usr/bin/env python
"""Single-rank RCCL smoke of every collective the N > 1 path of bench.py issues (the build box has one GPU: two ranks on one
device are refused by RCCL, so the multi-rank semantics are tested over gloo -- tests/test_distributed_cpu.py -- and this only
proves that the nccl(=RCCL) backend initialises here and accepts the calls, dtypes and stream usage):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 tools/rccl_smoke.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
w = dist.get_world_size()
t0 = time.perf_counter()
tables = torch.randn(100_001 * 128, device=dev)
dense = torch.randn(859_520, device=dev)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    work = dist.all_reduce(tables, op=dist.ReduceOp.SUM, async_op=True)          # GradReducer.reduce_tables_async
dist.all_reduce(dense, op=dist.ReduceOp.SUM)                                      # the dense bucket
work.wait()
torch.cuda.current_stream().wait_stream(side)
ids = torch.randint(0, 100_001, (20480,), device=dev)
rows = torch.randn(20480, 128, device=dev)
ids_all = torch.empty(w * ids.numel(), device=dev, dtype=ids.dtype)
rows_all = torch.empty((w * rows.shape[0], 128), device=dev)
dist.all_gather_into_tensor(ids_all, ids)                                         # SparseRowExchange.exchange
dist.all_gather_into_tensor(rows_all, rows)
assert torch.equal(ids_all[: ids.numel()], ids) and torch.equal(rows_all[: rows.shape[0]], rows)
acc = torch.ones(7, device=dev, dtype=torch.float64)
dist.all_reduce(acc)                                                              # compute_metrics
go = torch.ones(1, device=dev, dtype=torch.int32)
dist.all_reduce(go, op=dist.ReduceOp.MIN)                                         # the pre-heat agreement
tmax = torch.tensor([1.0], device=dev, dtype=torch.float64)
dist.all_reduce(tmax, op=dist.ReduceOp.MAX)                                       # timed_region
dist.barrier()
torch.cuda.synchronize()
print(f"rccl smoke ok: world {w}, backend {dist.get_backend()}, {time.perf_counter() - t0:.2f} s")
dist.destroy_process_group()
