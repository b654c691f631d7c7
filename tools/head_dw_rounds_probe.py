"""Is head_dw_split_kernel's launch two ROUNDS of its workgroups?  (review item 3b: 782 workgroups on 512 resident slots.)
Times the d W launch of the tied head at the benchmark's label-row count for item counts that give 512 / 782 / 1024 / 1536
workgroups of 128 items; if the launch is rounds of equal workgroups the times go 1 : 2 : 2 : 3, if it is work-bound
1 : 1.53 : 2 : 3.

    python tools/head_dw_rounds_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformers4rec_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)
N, D = 2780, 128
g = torch.Generator().manual_seed(0)
x = (0.5 * torch.randn(N, D, generator=g)).to(DEV)
for wgs in (256, 512, 640, 782, 1024, 1536):
    V = wgs * 128 if wgs != 782 else 100001
    W = (0.05 * torch.randn(V, D, generator=g)).to(DEV)
    y = torch.randint(1, V, (N,), generator=g).to(DEV)
    ws = ops.head_split_prepare(x, V)
    ldc = ops.logits_ld(V) if hasattr(ops, "logits_ld") else V
    logits, loss, rows, lse, dx = ops.head_split_logits_ce_dx(ws, x, W, y, ldc=ldc)
    dW = torch.zeros_like(W)
    for _ in range(3):
        ops.head_split_dw(ws, logits, lse, y, None, V, D, dW, accumulate=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.head_split_dw(ws, logits, lse, y, None, V, D, dW, accumulate=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{wgs:5d} workgroups (V {V}): {1e3 * ms:7.1f} us per d W call   {1e3 * ms / wgs:6.3f} us per workgroup")
    del W, dW, logits, ws
