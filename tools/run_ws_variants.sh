for v in "$@"; do
  echo "=== variant $v"
  UMNN_CC_LIB=$PWD/umnn_amd/libumnn_cc_${v}.so timeout 300 python tools/bwd_sweep.py --shape bsds300 --reps 3 2>&1 | grep -v amdgpu.ids | tail -9
done
