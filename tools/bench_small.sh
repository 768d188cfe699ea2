#!/bin/bash
# Small-ensemble / shard-size table (one GPU): cell-updates/s of bench.py at the ensemble sizes the lane mapping matters for.
#   tools/bench_small.sh <tag>      -> gpurun_out/small_<tag>.jsonl (one bench line per case)
set -e
tag=${1:-run}
out=gpurun_out/small_${tag}.jsonl
mkdir -p gpurun_out
: > $out
common="--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs"
for n in 1 2 8 32 64 128; do
  python bench.py --config c2 --nens $n $common >> $out
  echo "c2 nens=$n done"
done
python bench.py --config ref $common >> $out; echo "ref done"
python bench.py --config c3 $common >> $out; echo "c3 done"
python bench.py --config c4 $common >> $out; echo "c4 done"
python tools/show_small.py $out
