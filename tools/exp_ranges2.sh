#!/bin/bash
set -e
out=gpurun_out/exp_ranges_${1:-b}.jsonl
: > $out
common="--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs"
run() { echo "== $*" >> $out; python bench.py $common "$@" >> $out; echo "$* done"; }
run --config c2 --nens 128 --chunks 2 --indep 1 --span 32
run --config c2 --nens 128 --chunks 2 --indep 1 --span 64
run --config c2 --steps 2
run --config c2 --steps 2 --chunks 2 --indep 1
run --config c2 --steps 2 --chunks 4 --indep 1
run --config c3 --chunks 2 --indep 1 --span 32
run --config c3 --span 32
run --config c4 --chunks 2 --indep 1 --span 16
run --config c4 --span 16
run --config c4 --seg 4
python tools/show_small.py $out
