#!/bin/bash
# Regenerates the rocprofv3 evidence for config C2 on the GPU box (run through gpurun from the repo root):
#   gpurun_out/prof/  kernel stats of the default bench command, the bench JSON line printed under the profiler, PMC passes
#   (FETCH_SIZE, WRITE_SIZE, an SQ set, GRBM_GUI_ACTIVE; each in its own run, --kernel-trace only), their per-kernel summary
#   and the per-kernel HBM traffic JSON keyed by the content hash of pam_amd/csrc (bench.py reports `traffic` only when the
#   hash matches the build it runs).  Copy what should be judged into profiles/ with the round prefix.
#   The profiled command is `bench.py --chunks 1`: ONE member range, every stage kernel launched once per stage over the whole
#   ensemble -- the launches bench.py's `roofline` durations come from (its HIP-event pass also sets one range); the default schedule
#   runs the same kernels on two independent ranges of half the members each.
set -e
R=$PWD
OUT=$R/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
HASH=$(python3 -c "import sys; sys.path.insert(0,'$R'); import bench; print(bench.csrc_hash())")
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o t -- python3 $R/bench.py --chunks 1 --no-cpu-baseline --no-other-configs > $OUT/bench.log 2>&1 || { tail -20 $OUT/bench.log; exit 1; }
grep '^{"metric"' $OUT/bench.log > $OUT/bench_under_rocprof.json
cp $(find /tmp/ks -name '*kernel_stats.csv') $OUT/kernel_stats.csv
for ctr in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" GRBM_GUI_ACTIVE; do
  tag=$(echo $ctr | cut -d' ' -f1); [ "$tag" = SQ_INSTS_VALU ] && tag=SQ
  rm -rf /tmp/pmc_$tag
  timeout -k 10 500 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$tag -o t -- python3 $R/bench.py --chunks 1 --no-cpu-baseline --no-other-configs --no-kernel-timing --steps 1 --warmup 0 > $OUT/pmc_$tag.log 2>&1 || { tail -20 $OUT/pmc_$tag.log; exit 1; }
  cp $(find /tmp/pmc_$tag -name '*counter_collection.csv') $OUT/pmc_${tag}.csv
done
cd $R
PMC_SOURCE_CONFIG=c2 python3 tools/pmc_summary.py $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv $OUT/pmc_SQ.csv $OUT/pmc_GRBM_GUI_ACTIVE.csv --traffic-json $HASH > $OUT/pmc_summary.txt
tail -1 $OUT/pmc_summary.txt > $OUT/traffic.json
sed -i '$ d' $OUT/pmc_summary.txt
tail -12 $OUT/pmc_summary.txt
rm -f $OUT/*.log
