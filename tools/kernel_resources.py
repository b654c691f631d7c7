#!/usr/bin/env python
"""Registers, LDS, scratch and waves per SIMD of every kernel of the library, as the compiler reports them
(hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed):

    python tools/kernel_resources.py [substring ...]        # default: the kernels of the training step

Why it exists: head_fwd_dx_kernel was launched as "two workgroups per CU" for most of round 5 while the compiler had given it
213 + 80 registers = ONE wave per SIMD = one workgroup per CU; asked for two waves (launch_bounds(256, 2)) it fits 222 registers
without a spill and the step went 2.83 -> 2.70 ms.  The table this prints is where such a mismatch shows: a kernel whose
register count, not its LDS, caps the workgroups per CU below what the host code's launch geometry assumes."""
import glob
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformers4rec_amd.build import FLAGS, _hipcc  # noqa: E402

STEP = ["head_fwd_dx", "head_dw_split", "head_fdx_finalize", "split_w_images", "split_x_images", "xlnet_attn_block_fwd",
        "xlnet_ff_fwd", "xlnet_ff_bwd", "xlnet_ln1_bwd", "xlnet_dh_kernel", "xlnet_attn_mfma_bwd", "xlnet_proj_kernel",
        "gemm_f32_kernelILi64ELi64ELi16ELb1ELb0ELi0ELb1ELi0", "gemm_f32_kernelILi64ELi64ELi32ELb1ELb0ELi0ELb1ELi4",
        "layer_planes_tiled", "weight_scales", "seq_features_fwd_fast_kernelILi32ELi2ELi1ELb0", "emb_seg_sum_kernelILi2ELi1ELi16",
        "soft_embedding_bwd_kernelILi10", "adam_kernel", "dropout_kernel", "splitk_reduce"]


def one(src):
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([_hipcc()] + FLAGS + ["-c", src, "-o", os.path.join(d, "o.o"), "-Rpass-analysis=kernel-resource-usage"],
                           capture_output=True, text=True)
    out = []
    for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
        name = b.split()[0]
        g = lambda k: int(m.group(1)) if (m := re.search(k + r": (\d+)", b)) else -1      # noqa: E731
        out.append((os.path.basename(src), name, g("VGPRs"), g("AGPRs"), g(r"LDS Size \[bytes/block\]"),
                    g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
    return out


def main():
    want = sys.argv[1:] or STEP
    srcs = sorted(glob.glob(os.path.join(ROOT, "transformers4rec_amd", "csrc", "*.hip")))
    with ThreadPoolExecutor(8) as ex:
        rows = [r for rs in ex.map(one, srcs) for r in rs]
    print(f"{'file':24s} {'kernel (mangled, cut)':70s} {'VGPR':>5s} {'AGPR':>5s} {'LDS B':>7s} {'scratch':>7s} {'waves/SIMD':>10s}")
    for f, n, v, a, l, s, o in rows:
        if any(w in n for w in want):
            print(f"{f:24s} {n[:70]:70s} {v:5d} {a:5d} {l:7d} {s:7d} {o:10d}")
    print("(LDS B = static LDS only: the token-tile kernels take theirs at launch -- csrc/xlnet_fused*.hip, ~77-159 KB, one workgroup per CU)")


if __name__ == "__main__":
    main()
