#!/bin/bash
# kernel-trace statistics of the small-ensemble workloads (device-side kernel durations: what of a stage is launch gap?)
set -e
R=$PWD
OUT=$R/gpurun_out/prof_tiny
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "c2 1" "c2 8" "ref 1"; do
  set -- $spec
  rm -rf /tmp/kt_$1_$2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$1_$2 -o t -- python3 $R/bench.py --config $1 --nens $2 --steps 20 --warmup 2 --no-cpu-baseline --no-other-configs --no-kernel-timing > $OUT/$1_$2.log 2>&1 || { tail -20 $OUT/$1_$2.log; exit 1; }
  grep '^{"metric"' $OUT/$1_$2.log > $OUT/$1_nens$2_bench_under_rocprof.json
  cp $(find /tmp/kt_$1_$2 -name '*kernel_stats.csv') $OUT/$1_nens$2_kernel_stats.csv
  rm -f $OUT/$1_$2.log
  echo "== $1 nens=$2"; head -12 $OUT/$1_nens$2_kernel_stats.csv | cut -c1-160
done
