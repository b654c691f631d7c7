#!/usr/bin/env python
"""Where the HOST spends a training step of the bench workload (is the launch queue ahead of the GPU?):
per-step wall time of model() up to the label-count wait, the wait itself, the head, backward(), optimizer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from transformers4rec_amd import masking as mk

dev = torch.device("cuda", 0)
tr, schema, model, dense, tables, opt = bench.build(dev, 0.3)
reducer, _ = bench.setup_data_parallel(tr, model, dense, tables, 1)
batches = [tr.random_data_from_schema(schema, bench.BATCH, bench.SEQ, seed=i, device=dev) for i in range(4)]
model.train()
T = {}
orig = mk.MaskSequence.n_labels


def timed_n(self):
    t = time.perf_counter()
    r = orig(self)
    T["wait"] = T.get("wait", 0.0) + time.perf_counter() - t
    T["t_wait_end"] = time.perf_counter()
    return r


mk.MaskSequence.n_labels = timed_n
rows = []
for i in range(60):
    T.clear()
    t0 = time.perf_counter()
    out = model(batches[i % 4], training=True)
    t1 = time.perf_counter()
    out["loss"].backward()
    t2 = time.perf_counter()
    reducer.reduce_all()
    opt.step(grad_scale=reducer.grad_scale)
    t3 = time.perf_counter()
    if i >= 20:
        rows.append((t1 - t0, T.get("wait", 0.0), t1 - T.get("t_wait_end", t1), t2 - t1, t3 - t2, t3 - t0))
torch.cuda.synchronize()
import statistics as st
names = ("model() total", "  of which label-count wait", "  head after the wait", "backward()", "reduce + optimizer", "step (host)")
for k, n in enumerate(names):
    print(f"{n:30s} {1e3 * st.mean(r[k] for r in rows):7.3f} ms")
