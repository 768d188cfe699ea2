run() { lbl=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lbl', round(d['value']/1e9,4), 'G/s', round(d['ms_per_step'],1), 'ms', {k:round(v['avg_ms'],3) for k,v in d['kernels'].items() if k in ('flux','update','fct_mult')})"; }
run "default chunks=3" --chunks 3
for ch in 2 3 4 6; do
  PAMA_FLUX_HIGH=1 run "fluxhigh chunks=$ch" --chunks $ch
  PAMA_FLUX_HIGH=1 run "fluxhigh chunks=$ch floor0" --chunks $ch --lds-floor 0
done
run "default chunks=3 again" --chunks 3
