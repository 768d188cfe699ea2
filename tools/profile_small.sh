#!/bin/bash
# rocprofv3 evidence for the small-ensemble configs C3 and C4 (run through gpurun from the repo root), same passes as
# tools/profile_c2.sh:  gpurun_out/prof_small/c{3,4}_kernel_stats.csv, the bench JSON line printed under the profiler, PMC passes
# (FETCH_SIZE, WRITE_SIZE, an SQ set, GRBM_GUI_ACTIVE; each in its own run, --kernel-trace only), their per-kernel summary and
# the per-kernel HBM traffic JSON keyed by the content hash of pam_amd/csrc.
#   usage: tools/profile_small.sh [c3 c4 ...]
set -e
R=$PWD
OUT=$R/gpurun_out/prof_small
rm -rf $OUT && mkdir -p $OUT
HASH=$(python3 -c "import sys; sys.path.insert(0,'$R'); import bench; print(bench.csrc_hash())")
CFGS=${@:-c3 c4}
cd /tmp && export TMPDIR=/tmp
for cfg in $CFGS; do
  rm -rf /tmp/ks_$cfg
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$cfg -o t -- python3 $R/bench.py --config $cfg --chunks 1 --no-cpu-baseline --no-other-configs > $OUT/$cfg.log 2>&1 || { tail -20 $OUT/$cfg.log; exit 1; }
  grep '^{"metric"' $OUT/$cfg.log > $OUT/${cfg}_bench_under_rocprof.json
  cp $(find /tmp/ks_$cfg -name '*kernel_stats.csv') $OUT/${cfg}_kernel_stats.csv
  rm -f $OUT/$cfg.log
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" GRBM_GUI_ACTIVE; do
    tag=$(echo $ctr | cut -d' ' -f1); [ "$tag" = SQ_INSTS_VALU ] && tag=SQ
    rm -rf /tmp/pmc_${cfg}_$tag
    timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${cfg}_$tag -o t -- python3 $R/bench.py --config $cfg --chunks 1 --no-cpu-baseline --no-other-configs --no-kernel-timing --steps 1 --warmup 0 > $OUT/pmc_${cfg}_$tag.log 2>&1 || { tail -20 $OUT/pmc_${cfg}_$tag.log; exit 1; }
    cp $(find /tmp/pmc_${cfg}_$tag -name '*counter_collection.csv') $OUT/${cfg}_pmc_${tag}.csv
    rm -f $OUT/pmc_${cfg}_$tag.log
  done
  ( cd $R && PMC_SOURCE_CONFIG=$cfg python3 tools/pmc_summary.py $OUT/${cfg}_pmc_FETCH_SIZE.csv $OUT/${cfg}_pmc_WRITE_SIZE.csv $OUT/${cfg}_pmc_SQ.csv $OUT/${cfg}_pmc_GRBM_GUI_ACTIVE.csv --traffic-json $HASH > $OUT/${cfg}_pmc_summary.txt )
  tail -1 $OUT/${cfg}_pmc_summary.txt > $OUT/${cfg}_traffic.json
  sed -i '$ d' $OUT/${cfg}_pmc_summary.txt
  echo "== $cfg"; grep -E 'derived|wave-cycle' $OUT/${cfg}_pmc_summary.txt | grep -E 'flux|xupd|xtr|fct|ptail|trfix' || true
done
ls $OUT
