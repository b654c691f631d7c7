#!/usr/bin/env python
"""One process = one variant of the embedding gather (T4R_GATHER_U / T4R_GATHER_NT are read once per process) at the three
points bench.py's `roofline_gather` quotes, HIP-graph replay, plus the box's copy ceilings (torch copy_, tools/t4r_tools.hip
float4 kernel plain / non-temporal) and a random-row READ-ONLY probe (the gather without its write stream: what the HBM gives
random 512-byte rows).  Prints one JSON line.  tools/gather_sweep.sh loops over the variants."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from transformers4rec_amd import ops
import t4r_tools

dev = "cuda"


def graph_ms(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * reps)


GB, SEQ, D, rows = 8192, 20, 128, 10_000_001
res = {"U": os.environ.get("T4R_GATHER_U", "default"), "NT": os.environ.get("T4R_GATHER_NT", "auto")}
Wb = torch.empty((rows, D), device=dev).normal_()
ids = [torch.randint(1, rows, (GB, SEQ), device=dev) for _ in range(4)]
st = {"k": 0}
# (a) per-GPU batch, 100 001-row table (cache resident)
Ws = torch.empty((100_001, D), device=dev).normal_()
ids_s = torch.randint(1, 100_001, (1024, SEQ), device=dev)
f_s = [dict(kind=0, input=ids_s, table=Ws, dim=D, col=0, rows=100_001)]
ms = graph_ms(lambda: ops.seq_features_fwd(f_s, "concat", 1024, SEQ, SEQ, D), 50)
res["c2_per_gpu"] = {"us": round(ms * 1e3, 2), "frac": round(1024 * SEQ * (8 + 8 * D) / ms / 1e6 / 8000, 4)}
# (b) global batch, out of cache
feats = [[dict(kind=0, input=t, table=Wb, dim=D, col=0, rows=rows)] for t in ids]


def big():
    st["k"] += 1
    ops.seq_features_fwd(feats[st["k"] % 4], "concat", GB, SEQ, SEQ, D)


ms = graph_ms(big)
res["global_out_of_cache"] = {"us": round(ms * 1e3, 1), "frac": round(GB * SEQ * (8 + 8 * D) / ms / 1e6 / 8000, 4)}
# (c) C3 at the global batch
cats = [torch.empty((card, 64), device=dev).normal_() for card in (1001, 501, 101)]
dn = [torch.randn(GB * SEQ, 8, device=dev) for _ in range(2)]
fc3 = []
for t in ids:
    f = [dict(kind=0, input=t, table=Wb, dim=D, col=0, rows=rows)]
    col = D
    for tab in cats:
        f.append(dict(kind=0, input=t % tab.shape[0], table=tab, dim=64, col=col, rows=tab.shape[0])); col += 64
    for d_ in dn:
        f.append(dict(kind=1, input=d_, table=None, dim=8, col=col, rows=0)); col += 8
    fc3.append(f)
W3 = D + 3 * 64 + 16


def c3():
    st["k"] += 1
    ops.seq_features_fwd(fc3[st["k"] % 4], "concat", GB, SEQ, SEQ, W3)


ms = graph_ms(c3)
res["c3_global"] = {"us": round(ms * 1e3, 1), "frac": round(GB * SEQ * (32 + 8 * W3) / ms / 1e6 / 8000, 4)}
if os.environ.get("T4R_SWEEP_CEILINGS", "0") == "1":
    src = Wb.view(-1)[: 1 << 28]
    dst = torch.empty_like(src)
    nb = 2.0 * src.numel() * 4
    res["copy_torch_GBps"] = round(nb / graph_ms(lambda: dst.copy_(src), 10) / 1e6, 1)
    for blocks in (1024, 2048, 4096, 8192):
        for mode in (0, 1):
            res[f"copy_f4_{'nt' if mode else 'plain'}_{blocks}_GBps"] = round(
                nb / graph_ms(lambda: t4r_tools.copy(dst, src, mode=mode, blocks=blocks), 10) / 1e6, 1)
    # random-row read-only probe: sum of gathered rows via index_select into a small accumulate (torch): rows read, little written
    idx = ids[0].view(-1)
    out = torch.empty((idx.numel(), D), device=dev)
    res["torch_index_select_GBps"] = round(idx.numel() * (8 + 8 * D) / graph_ms(lambda: torch.index_select(Wb, 0, idx, out=out), 10) / 1e6, 1)
print(json.dumps(res), flush=True)
