#!/usr/bin/env python
"""Body GEMM shapes of BASELINE configs[1] through the token-stationary kernel (T4R_TOK_GEMM=1, default) or the
general kernel (T4R_TOK_GEMM=0), stand-alone, HIP-graph replay (the kernels are 5-50 us)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers4rec_amd import ops

T, D = 20480, 128
dev = "cuda"


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


def run(name, M, N, K, tb, **kw):
    A = torch.randn(M, K, device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    out = torch.empty(M, N, device=dev)
    extra = {}
    if kw.get("gelu"):
        extra = dict(bias=torch.randn(N, device=dev), epilogue=ops.EPI_BIAS_GELU, aux=torch.empty(M, N, device=dev),
                     drop=(0.3, 1, ops.dropout_ctr_hi(1, 1, ops.SITE_FF_ACT)))
    if kw.get("bias"):
        extra = dict(bias=torch.randn(N, device=dev), epilogue=ops.EPI_BIAS)
    us = graph_time(lambda: ops.gemm(A, B, False, tb, out=out, **extra))
    print(f"{name:28s} M={M:6d} N={N:4d} K={K:4d} {'NT' if tb else 'NN'} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s fp32-eq", flush=True)


print("T4R_TOK_GEMM =", os.environ.get("T4R_TOK_GEMM", "1"), " T4R_TOK_NBW =", os.environ.get("T4R_TOK_NBW", "auto"), flush=True)
run("q (one of three)", T, D, D, False)
run("k_r (per-session keys)", 2 * T, D, D, False)
run("o", T, D, D, True)
run("ff1 + gelu + dropout", T, 4 * D, D, True, gelu=True)
run("ff1 plain", T, 4 * D, D, True)
run("ff2 + bias", T, D, 4 * D, True, bias=True)
run("d ffact (NN, N=512)", T, 4 * D, D, False)
run("d h1 (NN, K=512)", T, D, 4 * D, False)
