#!/usr/bin/env python
"""One training step of the bench model from a fixed state; prints gradient fingerprints (run with
different kernel-variant env switches and compare: differences beyond atomic-order noise mean a race)."""
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
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    x = tr.random_data_from_schema(schema, bench.BATCH, bench.SEQ, seed=5, device=dev)
    model.input_features.masking._rng_offset = 0            # same mask and dropout draws every repetition
    model.transformer_block.transformer._drop_offset = 0
    dense.grad.zero_(); tables.grad.zero_()
    out = model(x, training=True)
    out["loss"].backward()
    torch.cuda.synchronize()
    d, t = dense.grad.double(), tables.grad.double()
    print(f"rep {rep} loss {float(out['loss'].detach()):.6f} dense sum {float(d.sum()):+.9e} abs {float(d.abs().sum()):.9e} "
          f"tables sum {float(t.sum()):+.9e} abs {float(t.abs().sum()):.9e}", flush=True)
