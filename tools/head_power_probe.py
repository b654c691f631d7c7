#!/usr/bin/env python
"""clock and power while one kernel family runs back to back for a few seconds (rocm-smi sampled from a thread):
is the head / the fused feed-forward kernel clock-limited by the power cap?
    python tools/head_power_probe.py"""
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from transformers4rec_amd import _lib, ops

dev = torch.device("cuda", 0)
N, V, D, T = 2780, 100001, 128, 20480
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(N, D, device=dev, generator=g)
W = torch.randn(V, D, device=dev, generator=g) * 0.3
labels = torch.randint(0, V, (N,), device=dev, generator=g)
gout = torch.tensor(1.0, device=dev)
ws = ops.head_split_prepare(x, V)
logits, _, _, lse = ops.head_split_logits_ce(ws, x, W, labels, ldc=ops.pad_ld(V))
dW = torch.zeros(V, D, device=dev)
dX = torch.empty(N, D, device=dev)
h1 = torch.randn(T, D, device=dev)
prm = [torch.randn(4 * D, D, device=dev) * 0.05, torch.randn(4 * D, device=dev) * 0.05, torch.randn(D, 4 * D, device=dev) * 0.05,
       torch.randn(D, device=dev) * 0.05, torch.ones(D, device=dev), torch.zeros(D, device=dev)]
planes = torch.empty(_lib.load().t4r_xlnet_ff_planes_floats(D), device=dev)
_lib.call("t4r_xlnet_ff_prepare", ops._stream(), prm[0].data_ptr(), prm[1].data_ptr(), prm[2].data_ptr(), D, planes.data_ptr())
big = torch.empty(256 << 20, device=dev, dtype=torch.float32)

cases = {
    "idle": lambda: time.sleep(0.01),
    "head logits+CE": lambda: ops.head_split_logits_ce(ws, x, W, labels, ldc=ops.pad_ld(V)),
    "head dW": lambda: ops.head_split_dw(ws, logits, lse, labels, gout, V, D, dW),
    "head dX": lambda: ops.head_split_dx(ws, logits, lse, labels, gout, V, W, out=dX),
    "fused FF forward": lambda: ops.xlnet_ff_fwd(h1, planes, prm[1], prm[3], prm[4], prm[5], 0.03, 0.3, 7, 11, 12),
    "HBM copy 1 GB": lambda: big[: 128 << 20].copy_(big[128 << 20:]),
}


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5)
            out.append(r.stdout)
        except Exception as e:  # noqa: BLE001
            out.append(f"ERR {e}")
        time.sleep(0.3)


for name, fn in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    th.start()
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 2.5:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    us = 1e3 * e0.elapsed_time(e1) / max(n, 1)
    last = [o for o in out if not o.startswith("ERR")][-2:] or out[-1:]
    print(f"== {name}: {us:8.1f} us per launch")
    for o in last[-1:]:
        lines = [ln for ln in o.strip().splitlines() if ln]
        print("   ", " | ".join(lines[:3])[:400])
