#!/usr/bin/env python
"""Workload for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ_*): a calibration copy of
known size followed by the GEMM shapes of the hot path (3 launches each; tools/pmc_table.py
averages the last 3 dispatches per kernel+grid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers4rec_amd import ops
T, D, V, NM = 20480, 128, 100001, 2779
# calibration: ops.dropout(p=0) is a plain 16-byte streaming copy: reads 1 GiB, writes 1 GiB
src = torch.randn(256 * 1024 * 1024, device="cuda")
dst = torch.empty_like(src)
for _ in range(3): ops.dropout(src, 0.0, 0, 0, out=dst)
torch.cuda.synchronize()
del src, dst
def run(M, N, K, ta, tb, **kw):
    A = torch.randn((K, M) if ta else (M, K), device="cuda"); B = torch.randn((N, K) if tb else (K, N), device="cuda")
    out = torch.empty((M, ops.pad_ld(N)), device="cuda")[:, :N]
    for _ in range(3): ops.gemm(A, B, ta, tb, out=out, **kw)
    torch.cuda.synchronize()
run(4096, 4096, 4096, False, True)
run(NM, V, D, False, True)            # head logits
run(V, D, NM, True, False)            # head dW
run(NM, D, V, False, False, splitk=-1)  # head dX
run(T, 4 * D, D, False, True)         # ff1
# the materialised head for d_model <= 128 (csrc/head_split.hip): prepare, logits + CE statistics, d X, d W
x = torch.randn(NM, D, device="cuda"); W = torch.randn(V, D, device="cuda") * 0.3
labels = torch.randint(0, V, (NM,), device="cuda"); g = torch.tensor(1.0, device="cuda")
dW = torch.zeros(V, D, device="cuda")
for _ in range(3):
    ws = ops.head_split_prepare(x, V)
    logits, loss, rows, lse = ops.head_split_logits_ce(ws, x, W, labels, ldc=ops.pad_ld(V))
    ops.head_split_dx(ws, logits, lse, labels, g, V, W)
    ops.head_split_dw(ws, logits, lse, labels, g, V, D, dW)
torch.cuda.synchronize()
# round 5: the ONE-PASS forward (logits + CE + d X: head_fwd_dx_kernel) and the d W that follows it -- the head's traffic per
# step is now these two launches + the small ones (table images, finalize), instead of logits + d X + d W
if ops.head_split_fdx_supported(D):
    for _ in range(3):
        ws = ops.head_split_prepare(x, V)
        logits, loss, rows, lse, dxu = ops.head_split_logits_ce_dx(ws, x, W, labels, ldc=ops.pad_ld(V))
        ops.head_split_dw(ws, logits, lse, labels, g, V, D, dW)
    torch.cuda.synchronize()

# round 3: the token-tile-stationary fused kernels of the XLNet layer at the benchmark size (csrc/xlnet_fused*.hip)
from transformers4rec_amd import _lib
g32 = lambda *s, std=0.05: torch.randn(*s, device="cuda") * std
n, dh = 4, D // 4
prm = [g32(D, n, dh), g32(D, n, dh), g32(D, n, dh), g32(D, n, dh), g32(D, n, dh), g32(n, dh), g32(n, dh), 1 + g32(D), g32(D),
       g32(4 * D, D), g32(4 * D), g32(D, 4 * D), g32(D), 1 + g32(D), g32(D)]
h = g32(T, D, std=1.0); dy = g32(T, D, std=1.0)
z = lambda k: torch.zeros(k, device="cuda")
for _ in range(3):
    planes = ops.xlnet_layer_prepare(prm, D)
    qkv = ops.xlnet_qkv_proj(h, planes)
    kr = ops.xlnet_kr_proj(g32(2048 * 20, D, std=1.0), planes)
    h1, ao, mean, rstd = ops.xlnet_oproj_ln(qkv[0].contiguous(), h, planes, prm[7], prm[8], 0.03, (0.3, 7, 11))
    hout, sv = ops.xlnet_ff_fwd(h1, planes, prm[10], prm[12], prm[13], prm[14], 0.03, 0.3, 7, 12, 13)
    ops.xlnet_ff_bwd(dy, h1, sv, prm[13], planes, z(D), z(D), z(D), z(4 * D), 0.3, 7, 12, 13)
    dh, dao, dav = ops.xlnet_ln1_bwd(dy, ao, h, mean, rstd, prm[7], planes, z(D), z(D), (0.3, 7, 11))
    ops.xlnet_dh_(qkv, planes, dh)
torch.cuda.synchronize()
