#!/bin/bash
# EXPERIMENT build of the library (-DT4R_EXPERIMENTAL): same C ABI plus t4r_xlnet_attn_block_bwd[_part_floats], with
#   * the measured-and-not-kept kernels of this directory compiled in, and
#   * every A/B / tuning / fallback-forcing environment switch LIVE (csrc/t4r_common.h: t4r_exp_getenv; the product build
#     compiles them out and reads only the six variables of INTEGRATION.md section 5).
# Every source is recompiled with the flag (round 6: the switches live in all of them).
#   bash tools/experimental/build_variant.sh && T4R_HIP_LIB=tools/bin/libt4r_hip_exp.so T4R_GATHER_U=4 python bench.py ...
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
CSRC=$ROOT/transformers4rec_amd/csrc
OUT=$ROOT/tools/bin
mkdir -p $OUT/exp_obj
rm -f $OUT/exp_obj/*.o
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DT4R_EXPERIMENTAL=1 -I$CSRC"
n=0
pids=()
for src in $CSRC/*.hip $ROOT/tools/experimental/wgrad_stream.hip $ROOT/tools/experimental/wgrad_units.hip; do
  extra=""; case "$(basename $src)" in head_split.hip|xlnet_fused.hip|xlnet_fused_attn.hip|xlnet_attn_block.hip) extra="-fno-honor-nans";; esac      # as transformers4rec_amd/build.py: EXTRA_FLAGS
  /opt/rocm/bin/hipcc $FLAGS $extra -c $src -o $OUT/exp_obj/$(basename ${src%.hip}).o & pids+=($!)
  n=$((n + 1))
  if [ $((n % 8)) -eq 0 ]; then for p in "${pids[@]}"; do wait $p; done; pids=(); fi
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libt4r_hip_exp.so $OUT/exp_obj/*.o
echo $OUT/libt4r_hip_exp.so
