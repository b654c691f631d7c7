#!/usr/bin/env python
"""does the head's time follow the vocabulary width, or the number of 128-column tiles per CU?  times the three
vocabulary-wide kernels of csrc/head_split.hip at widths around multiples of 256 tiles (N = 2780 label rows, D = 128).
    python tools/head_quant_probe.py [V ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from transformers4rec_amd import ops

dev = torch.device("cuda", 0)
N, D = 2780, 128
Vs = [int(a) for a in sys.argv[1:]] or [65536, 98304, 100001, 114688, 131072]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n


for V in Vs:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(N, D, device=dev, generator=g)
    W = torch.randn(V, D, device=dev, generator=g) * 0.3
    labels = torch.randint(0, V, (N,), device=dev, generator=g)
    gout = torch.tensor(1.0, device=dev)
    ws = ops.head_split_prepare(x, V)
    logits, _, _, lse = ops.head_split_logits_ce(ws, x, W, labels, ldc=ops.pad_ld(V))
    dW = torch.zeros(V, D, device=dev)
    dX = torch.empty(N, D, device=dev)
    t_f = timeit(lambda: ops.head_split_logits_ce(ws, x, W, labels, ldc=ops.pad_ld(V)))
    t_w = timeit(lambda: ops.head_split_dw(ws, logits, lse, labels, gout, V, D, dW))
    t_x = timeit(lambda: ops.head_split_dx(ws, logits, lse, labels, gout, V, W, out=dX))
    tiles = (V + 127) // 128
    print(f"V={V:7d} tiles={tiles:5d} ({tiles / 256:.2f}/CU)  fwd+CE {t_f:7.1f} us ({1e3 * t_f / tiles:6.1f} ns/tile)  "
          f"dW {t_w:7.1f} us ({1e3 * t_w / tiles:6.1f})  dX {t_x:7.1f} us ({1e3 * t_x / tiles:6.1f})", flush=True)
