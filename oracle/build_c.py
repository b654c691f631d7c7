"""TEST INFRASTRUCTURE ONLY.  Compiles the C part of the oracle (oracle/device_rng.c) with gcc into
oracle/_build/libt4r_oracle_rng.so (git-ignored, travels to the GPU box with the snapshot).

    python oracle/build_c.py [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "device_rng.c")
OUT = os.path.join(HERE, "_build", "libt4r_oracle_rng.so")


def build(force=False):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        r = subprocess.run(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", SRC, "-o", OUT], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("gcc failed (oracle/device_rng.c):\n" + r.stdout + r.stderr)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
