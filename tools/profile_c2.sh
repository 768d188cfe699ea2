#!/bin/bash
# Regenerates the rocprofv3 evidence for config C2 on the GPU box (run through gpurun from the repo root):
#   gpurun_out/prof/  kernel stats (default = chunked schedule, and --chunks 1 = kernels back to back), the bench JSON
#   lines printed under the profiler, PMC passes (FETCH_SIZE, WRITE_SIZE, SQ_*, GRBM_GUI_ACTIVE; each in its own run,
#   --kernel-trace only), and their per-kernel summary.  Copy what should be judged into profiles/.
set -e
R=$PWD
OUT=$R/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in default chunks1; do
  extra=""; [ $mode = chunks1 ] && extra="--chunks 1"
  rm -rf /tmp/ks_$mode
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$mode -o t -- python3 $R/bench.py --no-cpu-baseline $extra > $OUT/bench_$mode.log 2>&1 || { tail -20 $OUT/bench_$mode.log; exit 1; }
  grep '^{"metric"' $OUT/bench_$mode.log > $OUT/bench_under_rocprof_$mode.json
  cp $(find /tmp/ks_$mode -name '*kernel_stats.csv') $OUT/kernel_stats_$mode.csv
done
for ctr in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" GRBM_GUI_ACTIVE; do
  tag=$(echo $ctr | cut -d' ' -f1); [ "$tag" = SQ_INSTS_VALU ] && tag=SQ
  rm -rf /tmp/pmc_$tag
  timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$tag -o t -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --chunks 1 --steps 1 --warmup 0 > $OUT/pmc_$tag.log 2>&1 || { tail -20 $OUT/pmc_$tag.log; exit 1; }
  cp $(find /tmp/pmc_$tag -name '*counter_collection.csv') $OUT/pmc_${tag}_chunks1.csv
  [ $tag = GRBM_GUI_ACTIVE ] && cp $(find /tmp/pmc_$tag -name '*kernel_trace.csv') $OUT/pmc_GRBM_kernel_trace.csv
done
cd $R
python3 tools/pmc_summary.py $OUT/pmc_FETCH_SIZE_chunks1.csv $OUT/pmc_WRITE_SIZE_chunks1.csv $OUT/pmc_SQ_chunks1.csv $OUT/pmc_GRBM_GUI_ACTIVE_chunks1.csv --flux-json > $OUT/pmc_summary.txt
tail -3 $OUT/pmc_summary.txt
rm -f $OUT/*.log
