"""What does the dropout RNG cost inside the fused feed-forward kernels?  Times xlnet_ff_fwd / xlnet_ff_bwd at the benchmark's token
count with p = 0 and p = 0.3 (the Philox keys are evaluated in both directions: the masks are recomputed, not stored).

    python tools/ff_dropout_cost_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformers4rec_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)
T, D = 20480, 128
g = torch.Generator().manual_seed(0)
r = lambda *s: (0.1 * torch.randn(*s, generator=g)).to(DEV)
names = ops.XLNET_PARAM_ORDER if hasattr(ops, "XLNET_PARAM_ORDER") else None
dh = D // 4
params = [r(D, 4, dh), r(D, 4, dh), r(D, 4, dh), r(D, 4, dh), r(D, 4, dh), r(4, dh), r(4, dh), 1 + r(D), r(D), r(4 * D, D), r(4 * D),
          r(D, 4 * D), r(D), 1 + r(D), r(D)]
planes = ops.xlnet_layer_prepare(params, D)
h1, dy = r(T, D) * 10, r(T, D)
b1, w2b, gam, bet = params[10], params[12], params[13], params[14]


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for p in (0.0, 0.3):
    kw = dict(drop_p=p, seed=5, ctr_act=ops.dropout_ctr_hi(1, 0, ops.SITE_FF_ACT), ctr_out=ops.dropout_ctr_hi(1, 0, ops.SITE_FF_OUT))
    hout, saved = ops.xlnet_ff_fwd(h1, planes, b1, w2b, gam, bet, 0.03, **kw)
    tf = timed(lambda: ops.xlnet_ff_fwd(h1, planes, b1, w2b, gam, bet, 0.03, **kw))
    zs = [torch.zeros(D, device=DEV), torch.zeros(D, device=DEV), torch.zeros(D, device=DEV), torch.zeros(4 * D, device=DEV)]
    tb = timed(lambda: ops.xlnet_ff_bwd(dy, h1, saved, gam, planes, *zs, **kw))
    print(f"p = {p}: xlnet_ff_fwd {tf:6.1f} us   xlnet_ff_bwd {tb:6.1f} us   (includes the wrappers' allocations)")
