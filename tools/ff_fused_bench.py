#!/usr/bin/env python
"""stand-alone timing of the fused feed-forward kernels (csrc/xlnet_fused.hip) at the benchmark size
(T = 20 480 tokens, d_model 128): HIP-event loop, average us per launch, matrix-core rate against the fp32 peak.
    python tools/ff_fused_bench.py [T] [D] [drop_p] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformers4rec_amd import _lib, ops  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20480
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
p = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
dev = "cuda"
g = lambda *s, std=0.05: torch.randn(*s, device=dev) * std
h1, W1, b1, W2, b2 = g(T, D, std=1.0), g(4 * D, D), g(4 * D), g(D, 4 * D), g(D)
gam, bet = 1 + g(D), g(D)
ffpre, ffact = torch.empty(T, 4 * D, device=dev), torch.empty(T, 4 * D, device=dev)
ffout, hout = torch.empty(T, D, device=dev), torch.empty(T, D, device=dev)
mean, rstd = torch.empty(T, device=dev), torch.empty(T, device=dev)
dy, dh1, dfo, dpre = g(T, D, std=1.0), torch.empty(T, D, device=dev), torch.empty(T, D, device=dev), torch.empty(T, 4 * D, device=dev)
dg, db, db2, db1 = (torch.zeros(D, device=dev) for _ in range(3)) + (torch.zeros(4 * D, device=dev),) if False else (
    torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(4 * D, device=dev))
part = torch.empty(_lib.load().t4r_xlnet_ff_bwd_part_floats(T, D), device=dev)
planes = torch.empty(_lib.load().t4r_xlnet_ff_planes_floats(D), device=dev)
P = lambda t: t.data_ptr()
_lib.call("t4r_xlnet_ff_prepare", ops._stream(), P(W1), P(b1), P(W2), D, P(planes))


def fwd():
    _lib.call("t4r_xlnet_ff_fwd", ops._stream(), P(h1), P(planes), P(b1), P(b2), P(gam), P(bet), P(ffpre), P(ffact),
              P(ffout), P(mean), P(rstd), P(hout), T, D, 0.03, p, 7, 11, 12)


def bwd():
    _lib.call("t4r_xlnet_ff_bwd", ops._stream(), P(dy), P(ffout), P(h1), P(mean), P(rstd), P(gam), P(ffpre), P(planes),
              P(dh1), P(dfo), P(dpre), P(dg), P(db), P(db2), P(db1), P(part), T, D, p, 7, 11, 12)


def timed(fn):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


flops = 2.0 * T * D * 4 * D * 2
def prep():
    _lib.call("t4r_xlnet_ff_prepare", ops._stream(), P(W1), P(b1), P(W2), D, P(planes))


def fwd_infer():
    _lib.call("t4r_xlnet_ff_fwd", ops._stream(), P(h1), P(planes), P(b1), P(b2), P(gam), P(bet), None, None, None, None, None,
              P(hout), T, D, 0.03, 0.0, 7, 11, 12)


print(f"ff_prepare: {timed(prep):.1f} us")
print(f"ff_fwd inference form (nothing saved, no Philox): {timed(fwd_infer):.1f} us")
for name, fn in (("ff_fwd", fwd), ("ff_bwd(+2 reduces)", bwd)):
    us = timed(fn)
    print(f"{name}: {us:.1f} us  {flops / us / 1e6:.1f} TFLOP/s ({flops / us / 1e6 / 157.3:.2f} of the fp32 matrix peak)  T={T} D={D} p={p}")
