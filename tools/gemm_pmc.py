import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers4rec_amd import ops
T, D, V, NM = 20480, 128, 100001, 2765
def run(M, N, K, ta, tb, **kw):
    A = torch.randn((K, M) if ta else (M, K), device="cuda"); B = torch.randn((N, K) if tb else (K, N), device="cuda")
    out = torch.empty((M, ops.pad_ld(N)), device="cuda")[:, :N]
    for _ in range(3): ops.gemm(A, B, ta, tb, out=out, **kw)
    torch.cuda.synchronize()
run(4096, 4096, 4096, False, True)
run(NM, V, D, False, True)
run(V, D, NM, True, False)
run(T, 4 * D, D, False, True)
