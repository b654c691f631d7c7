#!/usr/bin/env python
"""Host-side cost of a training step: time to ENQUEUE n steps vs time for the GPU to finish them."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
tr, schema, model, dense, tables, opt = bench.build(dev, 0.3)
model.train()
batches = [tr.random_data_from_schema(schema, bench.BATCH, bench.SEQ, seed=i, device=dev) for i in range(8)]
def step(i):
    out = model(batches[i % 8], training=True); out["loss"].backward(); opt.step()
for i in range(30): step(i)
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
for i in range(n): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, gpu-complete {1e3 * (t2 - t0) / n:.3f} ms/step")
