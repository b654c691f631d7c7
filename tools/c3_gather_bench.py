#!/usr/bin/env python
"""C3 input block (BASELINE.json configs[2]): item 128 + 3 categoricals x 64 + 2 soft-embedding rows x 8 = 336-wide
concatenation.  Gather rate by HIP-graph replay, generic kernel (T4R_GATHER_U=0) vs the multi-chunk fast path.
bytes: T * (4 ids x 8 + 4 x 320 table rows + 2 x 4 x 8 dense rows read + 4 x 336 written)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from transformers4rec_amd import ops

dev = torch.device("cuda", 0)
L, REPS = 20, 20


def graph_time(fn):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REPS):
                fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * REPS)


cards, dims = [100_001, 1001, 101, 11], [128, 64, 64, 64]
tabs = [torch.randn(c, d, device=dev) for c, d in zip(cards, dims)]
big = torch.randn(10_000_001, 128, device=dev)
print(f"T4R_GATHER_U={os.environ.get('T4R_GATHER_U', '2 (default)')}")
for item_tab, label in ((tabs[0], "item table 100k rows"), (big, "item table 10M rows (out of cache)")):
    for B in (1024, 8192, 65536):
        T = B * L
        sets = []
        for k in range(4):
            feats, col = [], 0
            for t, c, d in zip([item_tab] + tabs[1:], [item_tab.shape[0]] + cards[1:], dims):
                feats.append(dict(kind=0, input=torch.randint(1, c, (B, L), device=dev), table=t, dim=d, col=col, rows=c))
                col += d
            for _ in range(2):
                feats.append(dict(kind=1, input=torch.randn(T, 8, device=dev), table=None, dim=8, col=col))
                col += 8
            sets.append(feats)
        st = {"k": 0}

        def fwd():
            st["k"] += 1
            ops.seq_features_fwd(sets[st["k"] % 4], "concat", B, L, L, 336)
        ms = graph_time(fwd)
        by = T * (4 * 8 + 4 * 320 + 2 * 4 * 8 + 4 * 336)
        print(f"{label:36s} T={T:8d}  {ms*1e3:8.1f} us  {by/ms/1e6:7.0f} GB/s ({by/ms/1e6/8000:5.1%} of 8 TB/s)  bytes {by}", flush=True)
