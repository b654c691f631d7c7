#!/usr/bin/env python
"""Eval head at C2 size: materialised scores + top-k vs the fused rank-of-target GEMM epilogue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers4rec_amd import ops
N, V, D = 1024, 100001, 128
x = torch.randn(N, D, device="cuda"); W = torch.randn(V, D, device="cuda") * 0.1
y = torch.randint(0, V, (N,), device="cuda")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def materialised():
    s = ops.gemm(x, W, False, True, ldc=ops.pad_ld(V)); return ops.topk(s, 20, V)
print(f"scores + top-20: {timeit(materialised):8.1f} us | fused rank of target: {timeit(lambda: ops.rank_of_target(x, W, y)):8.1f} us")
