#!/bin/bash
# experiments on the per-GPU workloads of the 8-GPU configurations: independent member ranges, spans
set -e
out=gpurun_out/exp_ranges_${1:-a}.jsonl
: > $out
common="--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs"
run() { echo "== $*" >> $out; python bench.py $common "$@" >> $out; echo "$* done"; }
run --config c4
run --config c4 --chunks 2 --indep 1
run --config c4 --chunks 4 --indep 1
run --config c4 --chunks 2
run --config c3
run --config c3 --chunks 2 --indep 1
run --config c3 --chunks 4 --indep 1
run --config c2 --nens 128
run --config c2 --nens 128 --chunks 2 --indep 1
run --config c2 --nens 128 --span 32
run --config c2 --nens 128 --span 8
run --config c2 --nens 256
run --config c2 --nens 256 --chunks 2 --indep 1
python tools/show_small.py $out
