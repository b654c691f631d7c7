#!/usr/bin/env python
"""csrc/head_split.hip against an fp64 reference and against the general GEMM path it replaces.

    python tools/head_split_bench.py            # accuracy on small + bench shapes, then timings at BASELINE configs[1]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from transformers4rec_amd import ops

dev = torch.device("cuda", 0)


def reference(x, W, labels, alpha, smooth, gout, logits):
    """fp64: logits, and the two backward products formed from the GIVEN fp32 logits (what the kernels consume)"""
    lg = alpha * (x.double() @ W.double().t())
    N, V = logits.shape
    p = torch.softmax(logits.double(), dim=1)
    onehot = torch.zeros_like(p)
    onehot[torch.arange(N, device=dev), labels] = 1.0
    G = (gout / N) * (p - (1 - smooth) * onehot - smooth / V)
    return lg, alpha * (G @ W.double()), alpha * (G.t() @ x.double())


def check(N, V, D, alpha=1.0, smooth=0.0, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(N, D, device=dev, generator=g)
    W = torch.randn(V, D, device=dev, generator=g) * 0.3
    labels = torch.randint(0, V, (N,), device=dev, generator=g)
    gout = torch.tensor(1.7, device=dev)
    ws = ops.head_split_prepare(x, V)
    logits = ops.head_split_logits(ws, x, W, alpha=alpha, ldc=ops.pad_ld(V))
    loss, _, lse = ops.softmax_ce_fwd(logits, labels, V, smooth)
    dW0 = torch.randn(V, D, device=dev, generator=g) * 1e-4
    dW = dW0.clone()
    ops.head_split_dw(ws, logits, lse, labels, gout, V, D, dW, alpha=alpha, label_smoothing=smooth, accumulate=True)
    dX = ops.head_split_dx(ws, logits, lse, labels, gout, V, W, alpha=alpha, label_smoothing=smooth)
    lg64, dX64, dW64 = reference(x, W, labels, alpha, smooth, 1.7, logits)
    lg2, loss2, rows2, lse2 = ops.head_split_logits_ce(ws, x, W, labels, alpha=alpha, label_smoothing=smooth, ldc=ops.pad_ld(V))
    lse64 = torch.logsumexp(lg2.double(), dim=1)
    e_ce = (float((lg2 - logits).abs().max()), float((lse2.double() - lse64).abs().max()),
            abs(float(loss2) - float(loss)), float((rows2 - _).abs().max()))
    print(f"    fused CE: |logits - separate| {e_ce[0]:.1e}  |lse - fp64| {e_ce[1]:.1e}  |loss - separate| {e_ce[2]:.1e}  rows {e_ce[3]:.1e}")
    e_l = float((logits.double() - lg64).abs().max() / lg64.abs().max())
    e_x = float((dX.double() - dX64).abs().max() / dX64.abs().max())
    e_w = float(((dW - dW0).double() - dW64).abs().max() / dW64.abs().max())
    # the general path on the same inputs
    with ops.precision("fp32"):
        lg_g = ops.gemm(x, W, False, True, alpha=alpha, ldc=ops.pad_ld(V))
        dX_g = ops.gemm_softmax_grad(logits, lse, labels, gout, V, W, False, alpha=alpha, label_smoothing=smooth, splitk=-1)
        dW_g = ops.gemm_softmax_grad(logits, lse, labels, gout, V, x, True, alpha=alpha, label_smoothing=smooth)
    f_l = float((lg_g.double() - lg64).abs().max() / lg64.abs().max())
    f_x = float((dX_g.double() - dX64).abs().max() / dX64.abs().max())
    f_w = float((dW_g.double() - dW64).abs().max() / dW64.abs().max())
    print(f"N={N:5d} V={V:6d} D={D:3d} alpha={alpha} eps={smooth}: rel err vs fp64  logits {e_l:.2e} (fp32 MFMA {f_l:.2e})"
          f"  dX {e_x:.2e} ({f_x:.2e})  dW {e_w:.2e} ({f_w:.2e})", flush=True)
    ok = e_l < 4 * max(f_l, 1e-7) and e_x < 4 * max(f_x, 1e-6) and e_w < 4 * max(f_w, 1e-6) and max(e_ce) < 2e-5
    return ok


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n


def bench(N=2780, V=100001, D=128):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(N, D, device=dev, generator=g)
    W = torch.randn(V, D, device=dev, generator=g) * 0.3
    labels = torch.randint(0, V, (N,), device=dev, generator=g)
    gout = torch.tensor(1.0, device=dev)
    ws = ops.head_split_prepare(x, V)
    logits = ops.head_split_logits(ws, x, W, ldc=ops.pad_ld(V))
    _, _, lse = ops.softmax_ce_fwd(logits, labels, V, 0.0)
    dW = torch.zeros(V, D, device=dev)
    dX = torch.empty(N, D, device=dev)
    fl = 2.0 * N * V * D
    rows = []
    rows.append(("prepare (cut X)", timeit(lambda: ops.head_split_prepare(x, V)), 0))
    rows.append(("logits  head_split", timeit(lambda: ops.head_split_logits(ws, x, W, ldc=ops.pad_ld(V))), fl))
    rows.append(("logits + CE fused", timeit(lambda: ops.head_split_logits_ce(ws, x, W, labels, ldc=ops.pad_ld(V))), fl))
    rows.append(("dW      head_split", timeit(lambda: ops.head_split_dw(ws, logits, lse, labels, gout, V, D, dW)), fl))
    rows.append(("dX      head_split", timeit(lambda: ops.head_split_dx(ws, logits, lse, labels, gout, V, W, out=dX)), fl))
    for mode in ("auto", "fp32"):
        with ops.precision(mode):
            rows.append((f"logits  general {mode}", timeit(lambda: ops.gemm(x, W, False, True, ldc=ops.pad_ld(V))), fl))
            rows.append((f"dW      general {mode}", timeit(lambda: ops.gemm_softmax_grad(logits, lse, labels, gout, V, x, True, out=dW, accumulate=True)), fl))
            rows.append((f"dX      general {mode}", timeit(lambda: ops.gemm_softmax_grad(logits, lse, labels, gout, V, W, False, splitk=-1)), fl))
    rows.append(("softmax_ce_fwd", timeit(lambda: ops.softmax_ce_fwd(logits, labels, V, 0.0)), 0))
    for name, us, f in rows:
        extra = f"  {f / us / 1e6:7.1f} TF/s fp32-equivalent, {6 * f / us / 1e6 / 2500:5.1%} of the bf16 peak (6 products)" if f else ""
        print(f"{name:28s} {us:9.1f} us{extra}", flush=True)


if __name__ == "__main__":
    ok = True
    for cfg in [(77, 1000, 64), (33, 257, 32), (130, 999, 96), (200, 5000, 128, 0.5, 0.1), (2780, 100001, 128)]:
        ok &= check(*cfg)
    print("ACCURACY", "OK" if ok else "FAILED", flush=True)
    bench()
