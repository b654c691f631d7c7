#!/usr/bin/env python
"""GPU time between the end of one training step (after the optimizer) and the first kernel of the next step's
transformer body = masking + embedding kernels + any idle time of the device at the step boundary (HIP events, no profiler)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
tr, schema, model, dense, tables, opt = bench.build(dev, 0.3)
reducer, _ = bench.setup_data_parallel(tr, model, dense, tables, 1)
batches = [tr.random_data_from_schema(schema, bench.BATCH, bench.SEQ, seed=i, device=dev) for i in range(4)]
model.train()
ev_body, ev_end, ev_head = [], [], []
model.transformer_block.register_forward_pre_hook(lambda m, a: (ev_body.append(torch.cuda.Event(enable_timing=True)), ev_body[-1].record())[0] and None)
model.transformer_block.register_forward_hook(lambda m, a, o: (ev_head.append(torch.cuda.Event(enable_timing=True)), ev_head[-1].record())[0] and None)
for i in range(80):
    out = model(batches[i % 4], training=True)
    out["loss"].backward()
    reducer.reduce_all()
    opt.step(grad_scale=reducer.grad_scale)
    e = torch.cuda.Event(enable_timing=True); e.record(); ev_end.append(e)
torch.cuda.synchronize()
import statistics as st
gap = [ev_end[i].elapsed_time(ev_body[i + 1]) for i in range(30, 79)]
body = [ev_body[i].elapsed_time(ev_head[i]) for i in range(30, 79)]
rest = [ev_head[i].elapsed_time(ev_end[i]) for i in range(30, 79)]
print(f"end of step -> body start {1e3 * st.mean(gap):7.1f} us | body forward {1e3 * st.mean(body):7.1f} us | head + backward + optimizer {1e3 * st.mean(rest):7.1f} us")
