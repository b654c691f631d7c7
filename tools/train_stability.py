#!/usr/bin/env python
"""Runs N training steps of the bench model and prints loss statistics (max / mean / last): a race
between streams shows up as loss spikes that the single-stream run does not have."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.manual_seed(0)
tr, schema, model, dense, tables, opt = bench.build(dev, 0.3)
model.input_features.masking.seed = 1
model.transformer_block.transformer.seed = 2
model.train()
batches = [tr.random_data_from_schema(schema, bench.BATCH, bench.SEQ, seed=i, device=dev) for i in range(8)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
losses = torch.zeros(n, device=dev)
for i in range(n):
    out = model(batches[i % 8], training=True)
    out["loss"].backward()
    opt.step()
    losses[i] = out["loss"].detach()
l = losses.cpu()
print(f"steps {n} first {l[0]:.4f} last {l[-1]:.4f} mean(last 100) {l[-100:].mean():.4f} max(after 20) {l[20:].max():.4f} "
      f"argmax {int(l[20:].argmax()) + 20} n>12.2 {int((l[20:] > 12.2).sum())}")
