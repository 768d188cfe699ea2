run() { lbl=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timing --steps 4 --warmup 1 "$@" 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lbl', round(d['value']/1e9,4), 'G/s', round(d['ms_per_step'],1), 'ms')"; }
export PAMA_FLUX_HIGH=1
for rep in 1 2; do
  for ch in 8 10 12 16; do run "c2 fluxhigh ch$ch" --chunks $ch; done
done
for fl in 0 81920; do run "c2 fluxhigh ch8 floor$fl" --chunks 8 --lds-floor $fl; done
