#!/usr/bin/env python
"""Workload for the rocprofv3 --pmc passes over the HBM-bound gather / scatter kernels (FETCH_SIZE,
WRITE_SIZE; tools/pmc_table.py averages the last 3 dispatches per kernel+grid).  A calibration copy of
known size first, then:
  A  C2 gather: 20 480 tokens, 100 001 x 128 table (cache resident after the first launch)
  B  global-batch gather: 163 840 tokens against a 10 000 001 x 128 table (5.1 GB), FRESH ids per launch
  C  C3 multi-feature gather (item 128 + 3 x 64 + 2 x 8 dense = 336 wide), 20 480 and 163 840 tokens
  D  table-gradient scatter, sorted form, C2 and 163 840 tokens (10 M-row table)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from transformers4rec_amd import ops

dev = "cuda"
L = 20
src = torch.randn(256 * 1024 * 1024, device=dev)
dst = torch.empty_like(src)
for _ in range(3):
    ops.dropout(src, 0.0, 0, 0, out=dst)       # p = 0: a plain 16-byte streaming copy, 1 GiB read + 1 GiB write
torch.cuda.synchronize()
del src, dst


def gather(table, B, fresh):
    V, D = table.shape
    for i in range(3):
        ids = torch.randint(1, V, (B, L), device=dev) if (fresh or i == 0) else ids
        ops.seq_features_fwd([dict(kind=0, input=ids, table=table, dim=D, col=0, rows=V)], "concat", B, L, L, D)
    torch.cuda.synchronize()


W_small = torch.randn(100_001, 128, device=dev)
gather(W_small, 1024, False)                              # A
W_big = torch.randn(10_000_001, 128, device=dev)
gather(W_big, 8192, True)                                 # B
for B in (1024, 8192):                                    # C
    cards, dims = [100_001, 1001, 101, 11], [128, 64, 64, 64]
    tabs = [W_small] + [torch.randn(c, d, device=dev) for c, d in zip(cards[1:], dims[1:])]
    for i in range(3):
        feats, col = [], 0
        for t, c, d in zip(tabs, cards, dims):
            feats.append(dict(kind=0, input=torch.randint(1, c, (B, L), device=dev), table=t, dim=d, col=col, rows=c))
            col += d
        for _ in range(2):
            feats.append(dict(kind=1, input=torch.randn(B * L, 8, device=dev), table=None, dim=8, col=col))
            col += 8
        ops.seq_features_fwd(feats, "concat", B, L, L, col)
    torch.cuda.synchronize()
for table, B in ((W_small, 1024), (W_big, 8192)):         # D
    V, D = table.shape
    dW = torch.zeros_like(table)
    for i in range(3):
        ids = torch.randint(1, V, (B, L), device=dev)
        dy = torch.randn(B * L, D, device=dev)
        keys, perm = ops.sort_ids(ids, V, 0)
        ops.embedding_bwd_sorted(dy, keys, perm, dW, 0, D)
    torch.cuda.synchronize()
    del dW
