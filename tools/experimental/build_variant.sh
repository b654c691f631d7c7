#!/bin/bash
# A/B variant of the product library with the measured-and-not-kept kernels of this directory compiled in
# (-DT4R_EXPERIMENTAL): same C ABI plus t4r_xlnet_attn_block_bwd[_part_floats]; the env switches of README.md select them.
#   bash tools/experimental/build_variant.sh && T4R_HIP_LIB=tools/bin/libt4r_hip_exp.so python bench.py ...
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
CSRC=$ROOT/transformers4rec_amd/csrc
LIB=$ROOT/transformers4rec_amd/lib
OUT=$ROOT/tools/bin
mkdir -p $OUT/exp_obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DT4R_EXPERIMENTAL=1 -I$CSRC"
pids=()
for f in xlnet_attn_block xlnet_layer xlnet_attn gemm_f32; do
  /opt/rocm/bin/hipcc $FLAGS -c $CSRC/$f.hip -o $OUT/exp_obj/$f.o & pids+=($!)
done
/opt/rocm/bin/hipcc $FLAGS -c $ROOT/tools/experimental/wgrad_stream.hip -o $OUT/exp_obj/wgrad_stream.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
objs=$(ls $LIB/*.o | grep -v -e "/xlnet_attn_block.o" -e "/xlnet_layer.o" -e "/xlnet_attn.o" -e "/gemm_f32.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libt4r_hip_exp.so $objs $OUT/exp_obj/*.o
echo $OUT/libt4r_hip_exp.so
