#!/bin/bash
# variant of the library whose attention block kernels leave s_memtime stamps at their phase borders (tools/attn_block_bench.py)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
LIB=$ROOT/transformers4rec_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $( [ "${1:-1}" != 0 ] && echo -DT4R_AB_STAMPS=${1:-1} ) ${2:+-DT4R_AB_STORE=$2} $T4R_EXTRA_FLAGS \
    -c $ROOT/transformers4rec_amd/csrc/xlnet_attn_block.hip -o /tmp/xlnet_attn_block_stamps.o
objs=$(ls $LIB/*.o | grep -v xlnet_attn_block.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LIB/libt4r_hip_${T4R_VARIANT:-stamps}.so $objs /tmp/xlnet_attn_block_stamps.o
echo $LIB/libt4r_hip_${T4R_VARIANT:-stamps}.so
