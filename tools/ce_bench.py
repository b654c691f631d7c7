#!/usr/bin/env python
"""softmax_ce_fwd_kernel on the C2 logits (HIP-graph replay).  T4R_HIP_LIB=<variant> for A/B."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers4rec_amd import ops
dev = torch.device("cuda", 0)
N, V = 2780, 100001
ld = ops.pad_ld(V)
buf = torch.randn(N, ld, device=dev)
lg = buf[:, :V]
y = torch.randint(1, V, (N,), device=dev)
keep = {}
def fn(): keep["r"] = ops.softmax_ce_fwd(lg, y, V, 0.0)
fn(); torch.cuda.synchronize()
s = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    fn()
    with torch.cuda.graph(g, stream=s):
        for _ in range(10): fn()
torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
ref = torch.nn.functional.cross_entropy(lg[:64].double(), y[:64], reduction="none")
err = float((keep["r"][1][:64].double() - ref).abs().max())
print(f"softmax_ce_fwd (+ mean) {us:7.1f} us  {N * V * 4 / us / 1e6:6.2f} TB/s  max |loss_row - fp64 ref| {err:.2e}")
