#!/bin/bash
# same-box A/B of backward-kernel builds: tools/run_ws_variants.sh <suffix...>  ("base" = the shipped library); two rounds
for round in 1 2; do
for v in "$@"; do
  lib=$PWD/umnn_amd/libumnn_cc_${v}.so; [ "$v" = base ] && lib=$PWD/umnn_amd/libumnn_cc.so
  echo "=== variant $v"
  UMNN_CC_LIB=$lib timeout 300 python tools/bwd_sweep.py --shape bsds300 --reps 5 2>&1 | grep -v amdgpu.ids | tail -1
done
done
