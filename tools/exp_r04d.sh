#!/bin/bash
set -e
out=gpurun_out/exp_r04d.jsonl
: > $out
common="--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs"
run() { echo "== $*" >> $out; python bench.py $common "$@" >> $out; echo "$* done"; }
run --config c4
run --config c3
run --config c2 --nens 128
run --config c2 --nens 64
run --config c2 --nens 256
run --config c2 --nens 512
run --config c2 --steps 2
python tools/show_small.py $out
