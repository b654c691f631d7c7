#!/bin/bash
# Stages a SCRATCH copy of the reference's two pure-Python packages under .scratch_ref/ (git-ignored, never committed)
# so that one gpurun call can run the reference's own Model / Head / SequentialBlock over the HIP drop-in modules on
# the GPU box, where /root/reference does not exist (tests/test_dropin_reference_gpu.py).  Remove with --clean.
#   tools/stage_reference.sh && gpurun -- 'python -m pytest tests/test_dropin_reference_gpu.py -m gpu -q'; tools/stage_reference.sh --clean
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
DST="$ROOT/.scratch_ref"
rm -rf "$DST"
[ "$1" = "--clean" ] && exit 0
SRC="${T4R_REFERENCE_SRC:-/root/reference}"
mkdir -p "$DST"
cp -r "$SRC/transformers4rec" "$SRC/merlin_standard_lib" "$DST/"
find "$DST" -name "__pycache__" -type d -prune -exec rm -rf {} +
echo "staged $(du -sh "$DST" | cut -f1) under $DST (git-ignored)"
