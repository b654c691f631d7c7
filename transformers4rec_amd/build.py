"""Builds libt4r_hip.so (all HIP kernels + the C ABI of include/t4r_hip.h) for gfx950 with
hipcc, in-tree (transformers4rec_amd/lib/).  hipcc cross-compiles without a GPU.

    python -m transformers4rec_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libt4r_hip.so")
SOURCES = ["gemm_f32.hip", "gemm_half.hip", "elementwise.hip", "embedding.hip", "masking.hip", "xlnet_attn.hip",
           "head.hip", "xlnet_layer.hip", "mha.hip", "swap_noise.hip", "xlnet_attn_mfma.hip", "mha_mfma.hip",
           "embedding_sorted.hip", "head_split.hip", "tok_gemm.hip", "embedding_bag.hip", "xlnet_fused.hip", "xlnet_fused_attn.hip", "xlnet_attn_block.hip",
           "xlnet_attn_long.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wno-unused-result"]
# per-source additions.  The head and the token-tile kernels: no NaN is ever looked at there (its maxima are over finite scores and -inf masks), and
# with NaNs honoured every fmaxf operand that comes out of a cross-lane move or a select is first canonicalised (v_max_f32 v, v, v:
# 31 of the one-pass head loop's 408 vector instructions); infinities stay honoured (-inf is the mask value).  Same bits.
EXTRA_FLAGS = {f: ["-fno-honor-nans"] for f in ("head_split.hip", "xlnet_fused.hip", "xlnet_fused_attn.hip", "xlnet_attn_block.hip")}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(s, []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def build_tools(force=False):
    """tools/t4r_tools.hip -> tools/bin/libt4r_tools.so: measurement infrastructure (a float4 copy kernel = the box's byte-moving
    ceiling, a CU occupier = what a resident RCCL kernel does to the chip).  NOT loaded by the package; bench.py and the tests
    load it through ctypes."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tools", "t4r_tools.hip")
    out_dir = os.path.join(root, "tools", "bin")
    out = os.path.join(out_dir, "libt4r_tools.so")
    os.makedirs(out_dir, exist_ok=True)
    if force or _stale(out, [src]):
        r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", src, "-o", out],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed (tools):\n" + r.stdout + r.stderr)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_tools(force="--force" in sys.argv))
