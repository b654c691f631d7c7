#!/usr/bin/env python
"""head d W (head_dw_split_kernel: one workgroup per 128 items, all label rows) timed at vocabulary sizes that give whole and
fractional residencies of the chip (2 workgroups per CU = 512 slots): is the 782-workgroup launch of BASELINE configs[1] paying
for two rounds (time ~ 2 x the 512-workgroup launch) or for its work (~ 1.53 x)?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from transformers4rec_amd import ops

N, D = 2765, 128
torch.manual_seed(0)
x = torch.randn(N, D, device="cuda")
g = torch.tensor(1.0, device="cuda")
for V in (32768, 65536, 81920, 98304, 100001, 114688, 131072, 196608):
    W = torch.randn(V, D, device="cuda") * 0.2
    y = torch.randint(0, V, (N,), device="cuda")
    ws = ops.head_split_prepare(x, V)
    logits, loss, rows, lse, dxu = ops.head_split_logits_ce_dx(ws, x, W, y, ldc=ops.pad_ld(V))
    dW = torch.zeros(V, D, device="cuda")
    for _ in range(5):
        ops.head_split_dw(ws, logits, lse, y, g, V, D, dW)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.head_split_dw(ws, logits, lse, y, g, V, D, dW)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    tiles = (V + 127) // 128
    print(f"V {V:7d} tiles {tiles:5d} = {tiles / 512:.2f} residencies   {us:7.1f} us   {us / tiles * 512:7.1f} us per 512 tiles")
    del W, logits, dW, ws
