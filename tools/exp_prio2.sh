run() { lbl=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timing --steps 4 --warmup 1 "$@" 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lbl', round(d['value']/1e9,4), 'G/s', round(d['ms_per_step'],1), 'ms')"; }
for rep in 1 2 3; do
  run "c2 default" 
  PAMA_FLUX_HIGH=1 run "c2 fluxhigh ch6" --chunks 6
  PAMA_FLUX_HIGH=1 run "c2 fluxhigh ch8" --chunks 8
  PAMA_FLUX_HIGH=1 run "c2 fluxhigh ch3" --chunks 3
done
for cfg in c3 c4; do
  run "$cfg default" --config $cfg
  for ch in 1 2 3 4; do PAMA_FLUX_HIGH=1 run "$cfg fluxhigh ch$ch" --config $cfg --chunks $ch;  run "$cfg updhigh ch$ch" --config $cfg --chunks $ch; done
done
