#!/usr/bin/env python
"""Diagnostic for bench.py's lockstep Recall@20 probe: |loss_hip - loss_oracle| along the joint training run
(same init, same masks, same Adam).  Round-off differences (1e-7) are amplified by Adam's m / sqrt(v) at a rate set by
the learning rate; this prints the trajectory so that a systematic difference would show as a jump, not a ramp."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch

import bench
import golden_utils as gu
import t4r_oracle as O

dev = torch.device("cuda", 0)
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 5e-3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 250
Vr = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
Br, Dr, NLr = 256, 64, 2
tr, schema, model, dense, tables, opt = bench.build(dev, 0.0, v_items=Vr, d_model=Dr, n_layer=NLr, n_head=4, lr=lr)
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
p = gu.oracle_params({"p/" + k: v.numpy() for k, v in sd.items()}, requires_grad=True)
leaves = [p["tables"]["item_id"], p["masked_item_embedding"]] + [t for lp in p["layers"] for t in lp.values()]
oopt = torch.optim.Adam(leaves, lr=lr)
cfg = dict(n_head=4, eps=0.03, item="item_id", masking="mlm")
active = 1 + torch.arange(Vr - 1)
masking = model.input_features.masking
model.train()
torch.set_num_threads(32)
for i in range(steps):
    ids = bench.markov_sessions(Br, bench.SEQ, active, 50_000 + i)
    out = model({"item_id": ids.to(dev)}, training=True)
    out["loss"].backward()
    opt.step()
    oopt.zero_grad()
    ref = O.session_forward(p, cfg, {"item_id": ids}, masking.mask_schema.cpu(), masking.masked_targets.cpu(), True, False)
    ref["loss"].backward()
    oopt.step()
    if i % 10 == 0 or i == steps - 1:
        a, b = float(out["loss"].detach()), float(ref["loss"].detach())
        wd = float((model.input_features.item_embedding_table.weight.detach().cpu() - p["tables"]["item_id"].detach()).abs().max())
        print(f"step {i:4d} loss hip {a:.6f} oracle {b:.6f} |d| {abs(a - b):.2e}  max |d table| {wd:.2e}", flush=True)
