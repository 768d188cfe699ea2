#!/bin/bash
# rocprofv3 kernel statistics of the small-ensemble configs C3 and C4 (run through gpurun from the repo root):
#   gpurun_out/prof_small/c{3,4}_kernel_stats.csv and the bench JSON line printed under the profiler.
set -e
R=$PWD
OUT=$R/gpurun_out/prof_small
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in c3 c4; do
  rm -rf /tmp/ks_$cfg
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$cfg -o t -- python3 $R/bench.py --config $cfg --no-cpu-baseline --no-other-configs > $OUT/$cfg.log 2>&1 || { tail -20 $OUT/$cfg.log; exit 1; }
  grep '^{"metric"' $OUT/$cfg.log > $OUT/${cfg}_bench_under_rocprof.json
  cp $(find /tmp/ks_$cfg -name '*kernel_stats.csv') $OUT/${cfg}_kernel_stats.csv
  rm -f $OUT/$cfg.log
done
ls $OUT
