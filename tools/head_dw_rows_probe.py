#!/usr/bin/env python
"""Per-ROW accuracy of the head's d W (one row = one item) against fp64, rows bucketed by their magnitude relative to the
largest row: what would the two-way fp16 split (T4R_HEAD_DW_FP16X2=1, an experiment switch) cost the rare items' rows,
next to the three bf16 planes (default) and the fp32 matrix cores (general GEMM)?  Also times the kernel.
    python tools/head_dw_rows_probe.py            # run once per setting of T4R_HEAD_DW_FP16X2"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from transformers4rec_amd import ops

dev = torch.device("cuda", 0)
N, V, D = 2780, 100001, 128
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(N, D, device=dev, generator=g)
# item popularity: row norms spread over two decades, so the softmax has a long tail of rare items
W = torch.randn(V, D, device=dev, generator=g) * (0.05 + 0.45 * torch.rand(V, 1, device=dev, generator=g) ** 3)
labels = torch.randint(0, V, (N,), device=dev, generator=g)
gout = torch.tensor(1.0, device=dev)
ws = ops.head_split_prepare(x, V)
logits, _, _, lse = ops.head_split_logits_ce(ws, x, W, labels, ldc=ops.pad_ld(V))
p = torch.softmax(logits.double(), dim=1)
p[torch.arange(N, device=dev), labels] -= 1.0
ref = (p / N).t() @ x.double()
del p
rmax = ref.abs().amax(dim=1)
gmax = float(rmax.max())


def report(name, dW):
    err = (dW.double() - ref).abs().amax(dim=1) / rmax.clamp_min(1e-300)
    print(f"{name}: norm-wise {float((dW.double() - ref).abs().max()) / gmax:.2e}")
    for lo, hi in ((1e-1, 1e1), (1e-2, 1e-1), (1e-3, 1e-2), (1e-4, 1e-3), (1e-5, 1e-4), (0.0, 1e-5)):
        m = (rmax / gmax >= lo) & (rmax / gmax < hi)
        if int(m.sum()):
            e = err[m]
            print(f"    rows with max |row| in [{lo:.0e}, {hi:.0e}) of the largest: {int(m.sum()):6d} rows, relative row error "
                  f"median {float(e.median()):.1e}  99 % {float(e.quantile(0.99)):.1e}  max {float(e.max()):.1e}")


dW = torch.zeros(V, D, device=dev)
ops.head_split_dw(ws, logits, lse, labels, gout, V, D, dW, accumulate=False)
report(f"head_split d W (T4R_HEAD_DW_FP16X2={os.environ.get('T4R_HEAD_DW_FP16X2', '0')})", dW)
with ops.precision("fp32"):
    dWg = ops.gemm_softmax_grad(logits, lse, labels, gout, V, x, True)
report("general GEMM, fp32 matrix cores", dWg)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    ops.head_split_dw(ws, logits, lse, labels, gout, V, D, dW, accumulate=False)
s.record()
for _ in range(20):
    ops.head_split_dw(ws, logits, lse, labels, gout, V, D, dW, accumulate=False)
e.record()
torch.cuda.synchronize()
print(f"head_split d W: {1e3 * s.elapsed_time(e) / 20:.1f} us per launch")
