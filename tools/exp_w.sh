#!/bin/bash
set -e
out=gpurun_out/exp_w.jsonl
: > $out
common="--steps 5 --warmup 2 --no-cpu-baseline --no-other-configs"
run() { echo "== $*" >> $out; python bench.py $common "$@" >> $out; echo "$* done"; }
run --config c2 --nens 32
run --config c2 --nens 32 --xtile 16,0,0
run --config c2 --nens 32 --xtile 8,0,0
run --config c2 --nens 16
run --config c2 --nens 16 --xtile 8,0,0
run --config c2 --nens 48
run --config c2 --nens 48 --xtile 16,0,0
run --config c2 --nens 48 --xtile 12,0,0
run --config c2 --nens 8
run --config c2 --nens 8 --xtile 0,0,2
run --config c2 --nens 8 --xtile 0,0,4
python tools/show_small.py $out
