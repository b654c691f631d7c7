#!/usr/bin/env python
"""configs[2] trained for a few steps from fixed seeds: the loss of every step at full precision and a checksum of the final
parameters -- run once per setting of a switch (T4R_EMB_SORT_MULTI, T4R_SOFT_EXACT, ...) to see WHERE two builds of the
step part ways (bit-identical steps give identical lines)."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
tr, schema, model, dense, tables, opt = bench.build(dev, 0.3, config="c3")
reducer, _ = bench.setup_data_parallel(tr, model, dense, tables, 1)
model.input_features.masking.seed, model.transformer_block.transformer.seed = bench.rank_seeds(0)
batches = [tr.random_data_from_schema(schema, bench.BATCH, bench.SEQ, seed=i, device=dev) for i in range(4)]
model.train()
step = bench.make_train_step(model, batches, reducer, opt)
losses = []
for i in range(steps):
    losses.append(float(step(i)["loss"]))
torch.cuda.synchronize()
h = hashlib.sha256()
for f in opt.flats:
    h.update(f.data.detach().cpu().numpy().tobytes())
print(" ".join(f"{x:.7f}" for x in losses))
print("params sha256", h.hexdigest()[:16])
