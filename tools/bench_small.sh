#!/bin/bash
# Small-ensemble / shard-size table (one GPU): cell-updates/s of bench.py at the ensemble sizes the lane mapping matters for.
#   tools/bench_small.sh <tag>      -> gpurun_out/small_<tag>.jsonl (one DETAIL object per case: bench.py --detail)
set -e
tag=${1:-run}
out=gpurun_out/small_${tag}.jsonl
mkdir -p gpurun_out
: > $out
common="--steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --detail gpurun_out/_small_detail.json"
for n in 1 2 8 32 64 128; do
  python bench.py --config c2 --nens $n $common > /dev/null 2>&1
  cat gpurun_out/_small_detail.json >> $out
  echo "c2 nens=$n done"
done
for cfg in ref c3 c4; do
  python bench.py --config $cfg $common > /dev/null 2>&1
  cat gpurun_out/_small_detail.json >> $out
  echo "$cfg done"
done
rm -f gpurun_out/_small_detail.json
python tools/show_small.py $out
