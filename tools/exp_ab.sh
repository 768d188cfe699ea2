# usage: tools/exp_ab.sh "<label>" [bench args...] -- runs the C2 bench in single-chunk and default mode
lbl=$1; shift
for ch in 1 0; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --chunks $ch --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lbl chunks=$ch', round(d['value']/1e9,4), 'G/s', round(d['ms_per_step'],1), 'ms', {k:round(v['avg_ms'],3) for k,v in d['kernels'].items() if k in ('flux','update','fct_mult')})"
done
