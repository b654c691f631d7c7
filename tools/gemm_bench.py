#!/usr/bin/env python
"""Micro-benchmark of t4r_gemm_f32 on the shapes the hot path launches (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers4rec_amd import ops

def bench(name, M, N, K, ta, tb, **kw):
    A = torch.randn((K, M) if ta else (M, K), device="cuda")
    B = torch.randn((N, K) if tb else (K, N), device="cuda")
    ld = ops.pad_ld(N)
    out = torch.empty((M, ld), device="cuda")[:, :N]
    if kw.get("pad_a"):   # A is a [*, :K or :M] slice of a padded buffer
        buf = torch.randn((A.shape[0], ops.pad_ld(A.shape[1])), device="cuda"); A = buf[:, :A.shape[1]]
        kw = {k: v for k, v in kw.items() if k != "pad_a"}
    for _ in range(3):
        ops.gemm(A, B, ta, tb, out=out, **kw)
    evs = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm(A, B, ta, tb, out=out, **kw); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
    tf = 2.0 * M * N * K / ms / 1e9
    print(f"{name:34s} M={M:6d} N={N:6d} K={K:6d} {'T' if ta else 'N'}{'T' if tb else 'N'} {ms*1e3:9.1f} us {tf:7.1f} TF/s", flush=True)

T, D, V, NM = 20480, 128, 100001, 2765
if len(sys.argv) > 1:
    ops.set_precision(sys.argv[1])
print("precision:", ops.get_precision(), flush=True)
bench("square 4096 NT", 4096, 4096, 4096, False, True)
bench("square 4096 NN", 4096, 4096, 4096, False, False)
bench("square 4096 TN", 4096, 4096, 4096, True, False)
bench("head logits", NM, V, D, False, True)
bench("head dW (acc)", V, D, NM, True, False, accumulate=True, pad_a=True)
bench("head dW (no acc)", V, D, NM, True, False, pad_a=True)
bench("head dX splitk", NM, D, V, False, False, splitk=-1, pad_a=True)
bench("qkv one", T, D, D, False, False)
bench("o proj", T, D, D, False, True)
bench("ff1", T, 4 * D, D, False, True)
bench("ff2", T, D, 4 * D, False, True)
bench("d_ffact (NN)", T, 4 * D, D, False, False)
bench("d_h1 (NN K=512)", T, D, 4 * D, False, False)
bench("wgrad w2 splitk", D, 4 * D, T, True, False, splitk=-1, accumulate=True)
bench("wgrad w1 splitk", 4 * D, D, T, True, False, splitk=-1, accumulate=True)
bench("wgrad o splitk", D, D, T, True, False, splitk=-1, accumulate=True)
bench("C5 body qkv T=102400 D=512", 102400, 512, 512, False, True)
bench("C5 body ff1 T=102400 512->3072", 102400, 3072, 512, False, True)
bench("C5 wgrad ff1", 3072, 512, 102400, True, False, splitk=-1, accumulate=True)
bench("C4 body ff1 T=51200 256->1024", 51200, 1024, 256, False, False)
if "--all" not in sys.argv:
    sys.exit(0)
# transposed-logits layout experiment: G^T stored [V, N_m]
bench("T-layout dW  NN M=V K=N_m", V, D, NM, False, False, pad_a=True)
bench("T-layout dX  TN K=V splitk", NM, D, V, True, False, splitk=-1, pad_a=True)
bench("T-layout logits^T NT M=V N=N_m", V, NM, D, False, True)
