#!/bin/bash
set -e
out=gpurun_out/exp_staged_${1:-a}.jsonl
: > $out
common="--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs"
run() { echo "== $*" >> $out; python bench.py $common "$@" >> $out; echo "$* done"; }
run --config c2 --nens 1
run --config c2 --nens 8
run --config c2 --nens 32
run --config ref
run --config c4 --xkernels tile --xtile 32,0,0
run --config c4 --xkernels tile --xtile 64,6,0
run --config c2 --nens 128 --xkernels tile --xtile 32,0,0
python tools/show_small.py $out
