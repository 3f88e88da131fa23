R=$PWD
run() { # dir label args...
  D=$1; L=$2; shift 2
  (cd $D && timeout 400 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$L', ' '.join(sys.argv[1:]), round(d['value']), round(d['ms_per_step'],4), round(r.get('avg_launch_ms',0),4))" "$@")
}
for i in 1 2; do
  for cfg in "--steps 20 --warmup 5" "--workload power --steps 100 --warmup 20" "--workload toy --steps 200 --warmup 40" "--workload vae --steps 100 --warmup 20" "--workload mnist --steps 50 --warmup 10" "--mode train --steps 10 --warmup 3" "--workload power --mode train --steps 20 --warmup 5" "--workload vae --mode train --steps 20 --warmup 5" "--workload mnist --mode train --steps 20 --warmup 5" "--workload mnist --mode train --embedding bf16 --steps 20 --warmup 5"; do
    run $R/_r05 r05 $cfg
    run $R r06 $cfg
  done
done
