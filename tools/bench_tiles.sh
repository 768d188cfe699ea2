#!/bin/bash
# x tile kernels against the sweeps on the per-GPU workloads of the 8-GPU configurations (one GPU)
set -e
tag=${1:-run}
out=gpurun_out/tiles_${tag}.jsonl
: > $out
common="--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs"
run() { echo "== $*" >> $out; python bench.py $common "$@" >> $out; echo "$* done"; }
run --config c4
run --config c4 --xkernels tile
run --config c4 --xkernels tile --xtile 64,6,0
run --config c4 --xkernels tile --xtile 32,0,0
run --config c4 --xkernels tile --xtile 16,0,0
run --config c3
run --config c3 --xkernels tile
run --config c3 --xkernels tile --xtile 64,6,0
run --config c3 --xkernels tile --xtile 32,0,0
run --config c2 --nens 128
run --config c2 --nens 128 --xkernels tile
run --config c2 --nens 128 --xkernels tile --xtile 64,6,0
run --config c2 --nens 128 --xkernels tile --xtile 32,0,0
run --config c2 --steps 2
run --config c2 --steps 2 --xkernels tile
run --config c2 --steps 2 --xkernels tile --xtile 32,0,0
python tools/show_small.py $out
