#!/usr/bin/env python
"""The attention block kernels (csrc/xlnet_attn_block.hip) at the BASELINE configs[1] shape, next to the launches they
replace; HIP-graph replay (no host overhead).  With the stamp variant of the library
(tools/build_stamp_variant.sh -> T4R_HIP_LIB=transformers4rec_amd/lib/libt4r_hip_stamps.so) it also prints the phase
durations in shader cycles (median over the waves of all workgroups).
    python tools/attn_block_bench.py [--once]        # --once: one launch of each kernel (for rocprofv3 --pmc passes)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

from transformers4rec_amd import _lib, ops
import exp_ops

dev = torch.device("cuda", 0)
B, L, n, D = 1024, 20, 4, 128
T = B * L
REPS = 1 if "--once" in sys.argv else 20
g = torch.Generator(device=dev).manual_seed(0)
ORDER = ("q", "k", "v", "o", "r", "r_w_bias", "r_r_bias", "ln_w", "ln_b", "w1", "b1", "w2", "b2", "ff_ln_w", "ff_ln_b")
r = lambda *s: 0.1 * torch.randn(*s, device=dev, generator=g)
dh = D // n
P = dict(q=r(D, n, dh), k=r(D, n, dh), v=r(D, n, dh), o=r(D, n, dh), r=r(D, n, dh), r_w_bias=r(n, dh), r_r_bias=r(n, dh),
         ln_w=1 + r(D), ln_b=r(D), w1=r(4 * D, D), b1=r(4 * D), w2=r(D, 4 * D), b2=r(D), ff_ln_w=1 + r(D), ff_ln_b=r(D))
planes = ops.xlnet_layer_prepare([P[k] for k in ORDER], D)
h = torch.randn(T, D, device=dev, generator=g)
dy = torch.randn(T, D, device=dev, generator=g)
lib = _lib.load()
stamps = None
if hasattr(lib, "t4r_debug_ab_stamps"):
    stamps = torch.zeros((B // 4 + 1) * 8 * 8, device=dev, dtype=torch.int64)
    lib.t4r_debug_ab_stamps.argtypes = [ctypes.c_void_p]
    lib.t4r_debug_ab_stamps(stamps.data_ptr())


def timed(fn):
    fn(); torch.cuda.synchronize()
    if REPS == 1:
        return 0.0
    s = torch.cuda.Stream(); gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(REPS): fn()
    torch.cuda.synchronize(); gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * REPS) * 1e3


for p in (0.3, 0.0):
    kr = torch.randn((B if p > 0 else 1) * 2 * L, D, device=dev, generator=g) * 0.3
    cp, co = ops.dropout_ctr_hi(1, 0, ops.SITE_PROB), ops.dropout_ctr_hi(1, 0, ops.SITE_ATTN_OUT)
    rw, rr = P["r_w_bias"].view(-1), P["r_r_bias"].view(-1)
    keep = {}

    def block_fwd():
        keep["h1"], keep["saved"] = ops.xlnet_attn_block_fwd(h, planes, P["o"].view(D, D), kr, rw, rr, P["ln_w"], P["ln_b"], B, L, n,
                                                             0.03, p, 7, cp, co)

    def chain_fwd():
        qkv = ops.xlnet_qkv_proj(h, planes)
        av, lse = ops.xlnet_attn_fwd(qkv[0], qkv[1], qkv[2], kr, rw, rr, B, L, n, drop=(p, 7, cp) if p > 0 else ops.NO_DROP)
        keep["c"] = ops.xlnet_oproj_ln(av, h, planes, P["ln_w"], P["ln_b"], 0.03, drop=(p, 7, co) if p > 0 else ops.NO_DROP)

    tb, tc = timed(block_fwd), timed(chain_fwd)
    print(f"dropout {p}: attention half forward: one kernel {tb:6.1f} us | projection + core + o-projection/LayerNorm launches {tc:6.1f} us", flush=True)
    if stamps is not None:
        stamps.zero_(); block_fwd(); torch.cuda.synchronize()
        st = stamps.view(-1, 8)[: (B // 4) * 8].double()
        d = st[:, 1:7] - st[:, 0:6]
        names = ["P (projection)", "barrier", "A (attention)", "barrier", "O products", "O epilogue"]
        if os.environ.get("T4R_AB_STAMP_DETAIL") == "P":
            names = ["h staging", "q, k products", "v products", "barrier + v stores", "-", "(unused)"]
            st[:, 5] = st[:, 4]
            st[:, 6] = st[:, 5]
        tot = (st[:, 6] - st[:, 0])
        print("   phase cycles, median over waves [min .. max]:")
        for i, nm in enumerate(names):
            print(f"      {nm:16s} {float(d[:, i].median()):9.0f}  [{float(d[:, i].min()):7.0f} .. {float(d[:, i].max()):7.0f}]")
        print(f"      {'total':16s} {float(tot.median()):9.0f}  [{float(tot.min()):7.0f} .. {float(tot.max()):7.0f}]")
    d_rw, d_rr, dg, db = (torch.zeros(D, device=dev) for _ in range(4))
    wq, wk, wv = (P[k].view(D, D) for k in ("q", "k", "v"))
    block_fwd()
    saved = keep["saved"]

    def block_bwd():
        keep["g"] = exp_ops.xlnet_attn_block_bwd(dy, saved, h, planes, wq, wk, wv, kr, rw, rr, P["ln_w"], d_rw, d_rr, dg, db, B, L, n, p, 7, cp, co)

    def chain_bwd():
        dh, dao, dav = ops.xlnet_ln1_bwd(dy, saved["ao"], h, saved["mean"], saved["rstd"], P["ln_w"], planes, dg, db,
                                         drop=(p, 7, co) if p > 0 else ops.NO_DROP)
        dq, dk, dv, dkr = ops.xlnet_attn_bwd(saved["qkv"][0], saved["qkv"][1], saved["qkv"][2], kr, rw, rr, saved["av"], saved["lse"], dav,
                                             d_rw, d_rr, B, L, n, drop=(p, 7, cp) if p > 0 else ops.NO_DROP)
        dqkv = torch.stack([dq, dk, dv])
        keep["c2"] = ops.xlnet_dh_(dqkv, planes, dh)

    tb, tc = timed(block_bwd), timed(chain_bwd)
    # the core alone: q | k | v as planes of one buffer (one-wave-per-head fp32-MFMA core) vs separate tensors (xlnet_attn_mfma.hip)
    dav_ = torch.randn(T, D, device=dev, generator=g)
    sep = [t.clone() for t in saved["qkv"]]
    core = lambda qs: (lambda: keep.__setitem__("core", ops.xlnet_attn_bwd(qs[0], qs[1], qs[2], kr, rw, rr, saved["av"], saved["lse"], dav_, d_rw, d_rr,
                                                                            B, L, n, drop=(p, 7, cp) if p > 0 else ops.NO_DROP)))
    print(f"dropout {p}: attention core backward alone: planes path {timed(core(saved['qkv'])):6.1f} us | separate-tensor path {timed(core(sep)):6.1f} us", flush=True)
    if stamps is not None and os.environ.get("T4R_AB_STAMP_DETAIL") == "C":       # the CORE launch (stamp level 3)
        stamps.zero_(); core(saved["qkv"])(); torch.cuda.synchronize()
        st = stamps.view(-1, 8)[: min(B, 512) * n].double()
        d = st[:, 1:8] - st[:, 0:7]
        print("   core launch, cycles, median over waves [min .. max]:")
        for i, nm in enumerate(["staging 1", "scores + gather (it 0)", "softmax bwd", "d v", "d q, d k, d k_r", "second query block", "rest (further sessions)"]):
            print(f"      {nm:24s} {float(d[:, i].median()):9.0f}  [{float(d[:, i].min()):7.0f} .. {float(d[:, i].max()):7.0f}]")
        print(f"      start spread over waves {float(st[:, 0].max() - st[:, 0].min()):9.0f}; first start -> last end {float(st[:, 7].max() - st[:, 0].min()):9.0f}")
    if stamps is not None and os.environ.get("T4R_AB_STAMP_DETAIL") == "3":
        stamps.zero_(); block_bwd(); torch.cuda.synchronize()
        st = stamps.view(-1, 8)[: (B // 4) * 8].double()
        d = st[:, 1:8] - st[:, 0:7]
        print("   backward phase 3 detail, cycles, median over waves [min .. max]:")
        for i, nm in enumerate(["staging 1", "scores + gather (it 0)", "softmax bwd", "d v", "d q, d k, d k_r", "second query block", "rest of phase 3"]):
            print(f"      {nm:24s} {float(d[:, i].median()):9.0f}  [{float(d[:, i].min()):7.0f} .. {float(d[:, i].max()):7.0f}]")
    elif stamps is not None and os.environ.get("T4R_AB_STAMP_DETAIL") != "P":
        stamps.zero_(); block_bwd(); torch.cuda.synchronize()
        st = stamps.view(-1, 8)[: (B // 4) * 8].double()
        d = st[:, 1:6] - st[:, 0:5]
        print("   backward phase cycles, median over waves [min .. max]:")
        for i, nm in enumerate(["1 LayerNorm bwd", "2 d attn_vec", "barrier", "3 attention core", "4 d h"]):
            print(f"      {nm:18s} {float(d[:, i].median()):9.0f}  [{float(d[:, i].min()):7.0f} .. {float(d[:, i].max()):7.0f}]")
        print(f"      {'total':18s} {float((st[:, 5] - st[:, 0]).median()):9.0f}")
    print(f"dropout {p}: attention half backward: one kernel (+ 2 partial reductions) {tb:6.1f} us | LayerNorm-1 backward + core + d h launches (+ stack) {tc:6.1f} us", flush=True)
