#!/usr/bin/env python
"""XLNet attention core (xlnet_attn_mfma_{fwd,bwd}) at the C2 shape, HIP-graph replay (no host overhead).
    T4R_HIP_LIB=<variant .so> python tools/attn_bench.py     # A/B of builds"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers4rec_amd import ops
dev = torch.device("cuda", 0)
B, L, n, D = 1024, 20, 4, 128
REPS = 20
g = torch.Generator(device=dev).manual_seed(0)
q, k, v, dout = (torch.randn(B * L, D, device=dev, generator=g) for _ in range(4))
for per_b in (True, False):
    kr = torch.randn((B if per_b else 1) * 2 * L, D, device=dev, generator=g)
    rw, rr = torch.randn(n, D // n, device=dev, generator=g), torch.randn(n, D // n, device=dev, generator=g)
    for p in (0.3, 0.0):
        drop = (p, 7, ops.dropout_ctr_hi(1, 0, ops.SITE_PROB)) if p > 0 else ops.NO_DROP
        keep = {}
        def fwd():
            keep["o"], keep["lse"] = ops.xlnet_attn_fwd(q, k, v, kr, rw, rr, B, L, n, drop)
        d_rw, d_rr = torch.zeros_like(rw), torch.zeros_like(rr)
        def bwd():
            keep["g"] = ops.xlnet_attn_bwd(q, k, v, kr, rw, rr, keep["o"], keep["lse"], dout, d_rw, d_rr, B, L, n, drop)
        res = []
        for fn in (fwd, bwd):
            fn(); torch.cuda.synchronize()
            s = torch.cuda.Stream(); gr = torch.cuda.CUDAGraph()
            with torch.cuda.stream(s):
                fn()
                with torch.cuda.graph(gr, stream=s):
                    for _ in range(REPS): fn()
            torch.cuda.synchronize(); gr.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); gr.replay(); e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / (2 * REPS) * 1e3)
        print(f"per-session k_r {per_b!s:5s} dropout {p}: fwd {res[0]:6.1f} us  bwd (+ partial reduce) {res[1]:6.1f} us", flush=True)
