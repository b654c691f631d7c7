#!/usr/bin/env python
"""The five weight-gradient launches of one XLNet layer backward at BASELINE configs[1] (q|k|v batched, o, r, w1, w2), stand-alone,
HIP-graph replay: what the second stream has to absorb per layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers4rec_amd import ops


def graph_time(fn, reps=20):
    fn(); torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


T, D, B, L = 20480, 128, 1024, 20
dev = "cuda"
h = torch.randn(T, D, device=dev); dqkv = torch.randn(3, T, D, device=dev); gq = torch.zeros(3, D, D, device=dev)
dao = torch.randn(T, D, device=dev); av = torch.randn(T, D, device=dev); go = torch.zeros(D, D, device=dev)
peb = torch.randn(2 * T, D, device=dev); dkr = torch.randn(2 * T, D, device=dev); gr = torch.zeros(D, D, device=dev)
dff = torch.randn(T, 4 * D, device=dev); h1 = torch.randn(T, D, device=dev); gw1 = torch.zeros(4 * D, D, device=dev)
dffo = torch.randn(T, D, device=dev); ffact = torch.randn(T, 4 * D, device=dev); gw2 = torch.zeros(D, 4 * D, device=dev)


def qkv():
    ops.call("t4r_gemm_f32", ops._stream(), 1, 0, D, D, T, 1.0, h.data_ptr(), D, dqkv.data_ptr(), D, gq.data_ptr(), D,
             None, 0, None, 0, -1, 1, 3, 0, T * D, D * D, 0.0, 0, 0)


launches = {
    "q|k|v (batched)": qkv,
    "o": lambda: ops.gemm(dao, av, True, False, splitk=-1, accumulate=True, out=go),
    "r (K = 2T)": lambda: ops.gemm(peb, dkr, True, False, splitk=-1, accumulate=True, out=gr),
    "w1 (512 x 128)": lambda: ops.gemm(dff, h1, True, False, splitk=-1, accumulate=True, out=gw1),
    "w2 (128 x 512)": lambda: ops.gemm(dffo, ffact, True, False, splitk=-1, accumulate=True, out=gw2),
}
tot = 0.0
for name, fn in launches.items():
    us = graph_time(fn)
    tot += us
    print(f"{name:20s} {us:7.1f} us", flush=True)
allf = lambda: [f() for f in launches.values()]
print(f"sum {tot:.1f} us; all five back to back {graph_time(allf):.1f} us; algorithmic 8.7 GFLOP per layer -> {8.72e9 / tot / 1e6:.1f} TF/s")

