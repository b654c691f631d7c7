#!/usr/bin/env python
"""Embedding gather (seq_features_fwd_kernel) and row-scatter backward (embedding_bwd_kernel) rate over
sizes, free of host launch overhead: N launches are captured into one HIP graph and replayed.
bytes: fwd T*(8 + 4D + 4D); bwd T*(8 + 4D) read + 2*U*4D RMW on U unique rows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers4rec_amd import ops

dev = torch.device("cuda", 0)
L = 20
REPS = 20


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


for V, D in [(100_001, 128), (10_000_001, 128), (10_000_001, 512)]:
    W = torch.randn(V, D, device=dev)
    for B in [1024, 8192, 65536, 262144]:
        T = B * L
        if T * D * 4 > 8e9:
            continue
        ids = torch.randint(1, V, (B, L), device=dev)
        feats = [dict(kind=0, input=ids, table=W, dim=D, col=0, rows=V)]
        out = {}
        def fwd():
            out["y"] = ops.seq_features_fwd(feats, "concat", B, L, L, D)
        ms = graph_time(fwd)
        by = T * (8 + 8 * D)
        line = f"V={V:>9d} D={D:3d} B={B:6d} T={T:8d}  fwd {ms*1e3:9.1f} us {by/ms/1e6:7.0f} GB/s ({by/ms/1e6/8000:5.1%})"
        dy = torch.randn(T, D, device=dev)
        dW = torch.zeros_like(W)
        def bwd():
            ops.embedding_bwd(dy, ids, dW, 0, D)
        ms = graph_time(bwd)
        U = int(torch.unique(ids).numel())
        by = T * (8 + 4 * D) + 2 * U * 4 * D
        line += f" | bwd {ms*1e3:9.1f} us {by/ms/1e6:7.0f} GB/s ({by/ms/1e6/8000:5.1%})"
        print(line, flush=True)
        del dy, dW
    del W
