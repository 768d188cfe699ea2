#!/bin/bash
set -e
out=gpurun_out/exp_tuning_${1:-b}.jsonl
: > $out
common="--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs"
run() { echo "== $*" >> $out; python bench.py $common "$@" >> $out; echo "$* done"; }
for cfg in c3 c4; do
  run --config $cfg
  run --config $cfg --tuning 3072,0,0
  run --config $cfg --tuning 3072,0,8192
  run --config $cfg --tuning 3072,8192,0
  run --config $cfg
done
python tools/show_small.py $out
