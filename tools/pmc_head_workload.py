#!/usr/bin/env python
"""Workload of bench.py's live HBM-traffic measurement (one rocprofv3 --pmc pass per counter; FETCH_SIZE and WRITE_SIZE do not
fit one pass: MI355X_MICROARCH.md "HBM"): a calibration copy of KNOWN size in the access pattern of the measured kernel
(16 bytes per lane, streaming: ops.dropout with p = 0 reads 1 GiB and writes 1 GiB), then three full calls of the one-pass head
forward (table maximum, table images, head_fwd_dx_kernel, finalize) at the label-row count given on the command line.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -o r -- python tools/pmc_head_workload.py N_ROWS
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from transformers4rec_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2780
V, D = 100_001, 128
src = torch.randn(256 * 1024 * 1024, device="cuda")
dst = torch.empty_like(src)
for _ in range(3):
    ops.dropout(src, 0.0, 0, 0, out=dst)
torch.cuda.synchronize()
del src, dst
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(N, D, device="cuda", generator=g)
W = torch.randn(V, D, device="cuda", generator=g) * 0.05
labels = torch.randint(1, V, (N,), device="cuda", generator=g)
gout = torch.tensor(1.0, device="cuda")
dW = torch.zeros(V, D, device="cuda")
for _ in range(3):
    ws = ops.head_split_prepare(x, V)
    logits, loss, rows, lse, dxu = ops.head_split_logits_ce_dx(ws, x, W, labels, ldc=ops.pad_ld(V))
    ops.head_split_dw(ws, logits, lse, labels, gout, V, D, dW)
torch.cuda.synchronize()
