#!/bin/bash
# Build an alternate libumnn_cc with extra -D flags on cc_backward_bf16.hip (A/B and timing builds):
#   tools/build_variant.sh <suffix> <flags...>   ->  umnn_amd/libumnn_cc_<suffix>.so   (select with UMNN_CC_LIB=...)
set -e
cd "$(dirname "$0")/.."
suffix=$1; shift
mkdir -p gpurun_out/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-sched-strategy=max-ilp "$@" \
    -c umnn_amd/csrc/cc_backward_bf16.hip -o /tmp/cc_backward_bf16_${suffix}.o
objs=$(ls umnn_amd/csrc/*.o | grep -v cc_backward_bf16.o)
hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/cc_backward_bf16_${suffix}.o -o umnn_amd/libumnn_cc_${suffix}.so
echo built umnn_amd/libumnn_cc_${suffix}.so
