#!/bin/bash
# Build an alternate libumnn_cc with extra -D flags on the fp16-piece forward (cc_forward_f16.hip) -- energy / timing probes
# (the probe switches live in tools/fwd_energy_probes.patch, not in the product header: `git apply tools/fwd_energy_probes.patch` first,
# `git checkout umnn_amd/csrc/cc_fwd_bf16_kernel.h` afterwards):
#   tools/build_fwd_variant.sh <suffix> <flags...>   ->  umnn_amd/libumnn_cc_<suffix>.so   (select with UMNN_CC_LIB=...)
set -e
cd "$(dirname "$0")/.."
suffix=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form "$@" \
    -c umnn_amd/csrc/cc_forward_f16.hip -o /tmp/cc_forward_f16_${suffix}.o
objs=$(ls umnn_amd/csrc/*.o | grep -v cc_forward_f16.o)
hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/cc_forward_f16_${suffix}.o -o umnn_amd/libumnn_cc_${suffix}.so
echo built umnn_amd/libumnn_cc_${suffix}.so
