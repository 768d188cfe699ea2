run() { lbl=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-timing --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lbl', round(d['value']/1e9,4), 'G/s', round(d['ms_per_step'],1), 'ms')"; }
export PAMA_FLUX_HIGH=1
for ne in 256 512; do for ch in 1 2 4 8; do [ $((ne/ch)) -ge 64 ] && run "c2 nens=$ne ch$ch" --nens $ne --chunks $ch; done; done
for ch in 8 16; do run "c2 nens=2048 ch$ch" --nens 2048 --chunks $ch; done
for ch in 1 2 3 4; do run "c3 nens=8192 ch$ch" --config c3 --nens 8192 --chunks $ch; done
