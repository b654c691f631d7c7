#!/usr/bin/env python
"""Workload for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ_*): a calibration copy of
known size followed by the GEMM shapes of the hot path (3 launches each; tools/pmc_table.py
averages the last 3 dispatches per kernel+grid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers4rec_amd import ops
T, D, V, NM = 20480, 128, 100001, 2779
# calibration: ops.dropout(p=0) is a plain 16-byte streaming copy: reads 1 GiB, writes 1 GiB
src = torch.randn(256 * 1024 * 1024, device="cuda")
dst = torch.empty_like(src)
for _ in range(3): ops.dropout(src, 0.0, 0, 0, out=dst)
torch.cuda.synchronize()
del src, dst
def run(M, N, K, ta, tb, **kw):
    A = torch.randn((K, M) if ta else (M, K), device="cuda"); B = torch.randn((N, K) if tb else (K, N), device="cuda")
    out = torch.empty((M, ops.pad_ld(N)), device="cuda")[:, :N]
    for _ in range(3): ops.gemm(A, B, ta, tb, out=out, **kw)
    torch.cuda.synchronize()
run(4096, 4096, 4096, False, True)
run(NM, V, D, False, True)            # head logits
run(V, D, NM, True, False)            # head dW
run(NM, D, V, False, False, splitk=-1)  # head dX
run(T, 4 * D, D, False, True)         # ff1
