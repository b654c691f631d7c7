#!/usr/bin/env python
"""the embedding gather at the point bench.py quotes as `at_global_batch_8192_out_of_cache`: 163 840 tokens against a
10 000 001 x 128 table (5.1 GB), FOUR id sets rotated so that neither L2 nor the 256 MB Infinity Cache can hold the rows
(one id set = 84 MB of rows), HIP-graph replay.  T4R_GATHER_U selects the tokens per lane group."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers4rec_amd import ops

dev = "cuda"
GB, SEQ, D, rows = 8192, 20, 128, 10_000_001
NSET = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W = torch.empty((rows, D), device=dev).normal_()
ids = [torch.randint(1, rows, (GB, SEQ), device=dev) for _ in range(NSET)]
feats = [[dict(kind=0, input=t, table=W, dim=D, col=0, rows=rows)] for t in ids]
st = {"k": 0}


def fn():
    st["k"] += 1
    ops.seq_features_fwd(feats[st["k"] % NSET], "concat", GB, SEQ, SEQ, D)


fn(); torch.cuda.synchronize()
s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
reps = 20
with torch.cuda.stream(s):
    fn()
    with torch.cuda.graph(g, stream=s):
        for _ in range(reps):
            fn()
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / (2 * reps)
b = GB * SEQ * (8 + 8 * D)
print(f"U={os.environ.get('T4R_GATHER_U', 'default')} sets={NSET}: {ms * 1e3:.1f} us  {b / ms / 1e6:.0f} GB/s  {b / ms / 1e6 / 8000:.3f} of 8 TB/s")
