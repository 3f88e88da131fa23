R=$PWD
for i in 1 2 3; do
  for L in libumnn_cc_prev.so libumnn_cc_expc.so libumnn_cc.so; do
    UMNN_CC_LIB=$R/umnn_amd/$L timeout 300 python $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 "$@" 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$L', round(d['value']), round(d['ms_per_step'],3), round(r.get('avg_launch_ms',0),4), r.get('sclk_mhz'), r.get('power_w'))"
  done
done
