#!/bin/bash
# rocprofv3 evidence for ONE bench.py workload (run through gpurun from the repo root):
#   tools/profile.sh <tag> [--quick] <bench.py arguments ...>      e.g.  tools/profile.sh c2 --chunks 1
#                                                                         tools/profile.sh c2grid_nens1 --config c2 --nens 1
#   -> gpurun_out/prof_<tag>/  kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py <args>`), bench_under_rocprof.json (the
#      line printed under the profiler) + bench_detail_under_rocprof.json, the PMC passes -- FETCH_SIZE, WRITE_SIZE, an SQ set,
#      GRBM_GUI_ACTIVE and a second SQ set (LDS / instruction-fetch / memory-instruction counters), each in its own run with
#      --kernel-trace only -- their per-kernel summary (tools/pmc_summary.py) and traffic.json, the per-kernel HBM traffic keyed by the
#      content hash of pam_amd/csrc (bench.py reports `traffic` only when the hash matches the build it runs).
# Copy what should be judged into profiles/ with the round prefix (tools/collect_profiles.sh).  `--chunks 1` profiles ONE member range:
# every stage kernel launched once per stage over the whole ensemble -- the launches bench.py's `roofline` durations come from.
# (This script replaces round 1-4's profile_c2.sh / profile_small.sh / profile_mid.sh / profile_tiny.sh.)
set -e
R=$PWD
tag=$1; shift
quick=0
if [ "$1" = "--quick" ]; then quick=1; shift; fi
OUT=$R/gpurun_out/prof_$tag
rm -rf $OUT && mkdir -p $OUT
HASH=$(python3 -c "import sys; sys.path.insert(0,'$R'); import bench; print(bench.csrc_hash())")
# the launch shapes of the small-ensemble kernels follow the device's CU count: the profile is valid for this build ON this kind of part
export PMC_COMPUTE_UNITS=$(python3 -c "import torch; print(torch.cuda.get_device_properties(0).multi_processor_count)")
common="--no-cpu-baseline --no-other-configs"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$tag
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -o t -- python3 $R/bench.py "$@" $common --detail $OUT/bench_detail_under_rocprof.json > $OUT/bench.log 2> $OUT/bench.err || { echo "kernel-trace pass failed:"; tail -20 $OUT/bench.err; tail -5 $OUT/bench.log; exit 1; }
rm -f $OUT/bench.err
grep '^{"metric"' $OUT/bench.log > $OUT/bench_under_rocprof.json
cp $(find /tmp/ks_$tag -name '*kernel_stats.csv') $OUT/kernel_stats.csv
rm -f $OUT/bench.log
SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQ2="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH SQ_INSTS_SMEM"
passes=("FETCH_SIZE" "WRITE_SIZE" "$SQ1" "GRBM_GUI_ACTIVE")
[ $quick = 0 ] && passes+=("$SQ2")
files=""
for ctr in "${passes[@]}"; do
  t=$(echo $ctr | cut -d' ' -f1); [ "$t" = SQ_INSTS_VALU ] && t=SQ; [ "$t" = SQ_INSTS_LDS ] && t=SQ2
  rm -rf /tmp/pmc_${tag}_$t
  if ! timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$t -o t -- python3 $R/bench.py "$@" $common --no-kernel-timing --steps 1 --warmup 0 --detail $OUT/unused.json > $OUT/pmc_$t.log 2>&1; then
    echo "PMC pass $t failed:"; tail -5 $OUT/pmc_$t.log
    [ "$t" = SQ2 ] && continue      # the second SQ set is an extra: a counter this rocprofv3 does not know must not lose the rest
    exit 1
  fi
  cp $(find /tmp/pmc_${tag}_$t -name '*counter_collection.csv') $OUT/pmc_$t.csv
  files="$files $OUT/pmc_$t.csv"
  rm -f $OUT/pmc_$t.log
done
rm -f $OUT/unused.json
cd $R
PMC_SOURCE_CONFIG=$tag python3 tools/pmc_summary.py $files --traffic-json $HASH > $OUT/pmc_summary.txt
tail -1 $OUT/pmc_summary.txt > $OUT/traffic.json
sed -i '$ d' $OUT/pmc_summary.txt
echo "== $tag"; grep -E 'derived|wave-cycle|per wavefront' $OUT/pmc_summary.txt | grep -E 'flux|xupd|xtr|fct|ptail|trfix' || true
