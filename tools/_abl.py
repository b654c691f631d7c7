import os, sys
sys.path.insert(0, "/root/repo")
import torch
from transformers4rec_amd import ops
from tools.head_split_bench import timeit
dev = torch.device("cuda", 0)
N, V, D = 2780, 100001, 128
x = torch.randn(N, D, device=dev); W = torch.randn(V, D, device=dev) * 0.3
ws = ops.head_split_prepare(x, V)
print(os.environ.get("T4R_HEAD_DBG"), os.environ.get("T4R_HEAD_ROWS_PER_WG"), "logits %.1f us" % timeit(lambda: ops.head_split_logits(ws, x, W, ldc=ops.pad_ld(V))))
