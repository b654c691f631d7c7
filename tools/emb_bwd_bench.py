#!/usr/bin/env python
"""Table-gradient scatter: fp32 row atomics (embedding_bwd_kernel) vs sort + segmented sum
(embedding_sorted.hip), HIP-graph replay (no host launch overhead).
bytes (SURVEY 8d, K1 bwd): T*(8 id + 4D grad row) + 2*U*4D RMW on the U unique rows; the sorted form
reads 8 B of (key, perm) per lookup instead of the id; the sort itself (ids only) is timed apart, it
runs in the forward pass."""
import os
import sys

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


def zipf_ids(V, shape):
    u = torch.rand(shape, device=dev, dtype=torch.float64)
    return (torch.exp(u * torch.log(torch.tensor(float(V), dtype=torch.float64))).long().clamp_(1, V - 1))


print("V D T dist | atomic us (frac of 8 TB/s) | sorted us (frac) | sort us")
for V, D in [(100_001, 128), (1001, 64), (11, 64), (1_000_001, 256), (10_000_001, 512)]:
    dW = torch.zeros(V, D, device=dev)
    for B in [1024, 8192, 65536]:
        T = B * L
        if T * D * 4 > 4e9:
            continue
        for dist in ("uniform", "zipf"):
            ids = torch.randint(1, V, (B, L), device=dev) if dist == "uniform" else zipf_ids(V, (B, L))
            dy = torch.randn(T, D, device=dev)
            U = int(torch.unique(ids).numel())
            by = T * (8 + 4 * D) + 2 * U * 4 * D
            ms_a = graph_time(lambda: ops.embedding_bwd(dy, ids, dW, 0, D))
            keys, perm = ops.sort_ids(ids, V, 0)
            ms_s = graph_time(lambda: ops.embedding_bwd_sorted(dy, keys, perm, dW, 0, D))
            ms_sort = graph_time(lambda: ops.sort_ids(ids, V, 0))
            print(f"V={V:>9d} D={D:3d} T={T:8d} {dist:7s} U={U:8d} | atomic {ms_a*1e3:8.1f} us ({by/ms_a/1e6/8000:5.1%}) | "
                  f"sorted {ms_s*1e3:8.1f} us ({by/ms_s/1e6/8000:5.1%}) | sort {ms_sort*1e3:7.1f} us", flush=True)
            del dy
    del dW
