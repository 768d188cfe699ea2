#!/bin/bash
# ragged ensembles (nens not a multiple of 64): member lanes against flat lanes + tile kernels
set -e
out=gpurun_out/exp_ragged_${1:-a}.jsonl
: > $out
common="--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs"
run() { echo "== $*" >> $out; python bench.py $common "$@" >> $out; echo "$* done"; }
for n in 48 70 96; do
  run --config c2 --nens $n
  run --config c2 --nens $n --lanes flat --xkernels tile
  run --config c2 --nens $n --lanes flat --xkernels sweep
done
run --config c2 --nens 70 --lanes member --xkernels tile
python tools/show_small.py $out
