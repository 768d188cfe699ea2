#!/bin/bash
set -e
out=gpurun_out/exp_graph_${1:-a}.jsonl
: > $out
common="--steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-kernel-timing"
run() { echo "== $*" >> $out; python bench.py $common "$@" >> $out; echo "$* done"; }
for g in off on; do
  run --config c2 --nens 1 --graph $g
  run --config c2 --nens 2 --graph $g
  run --config c2 --nens 8 --graph $g
  run --config ref --graph $g
done
python tools/show_small.py $out
