#!/bin/bash
# PMC counters of the small-ensemble kernels at a mid-size ensemble (default: C2 grid, 32 members)
set -e
R=$PWD
N=${1:-32}
OUT=$R/gpurun_out/prof_mid
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ctr in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE; do
  tag=$(echo $ctr | cut -d' ' -f1)
  rm -rf /tmp/pm_$tag
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pm_$tag -o t -- python3 $R/bench.py --config c2 --nens $N --no-cpu-baseline --no-other-configs --no-kernel-timing --steps 1 --warmup 0 > $OUT/$tag.log 2>&1 || { tail -20 $OUT/$tag.log; exit 1; }
  cp $(find /tmp/pm_$tag -name '*counter_collection.csv') $OUT/pmc_${tag}.csv
  rm -f $OUT/$tag.log
done
cd $R
python3 tools/pmc_summary.py $OUT/pmc_*.csv > $OUT/pmc_summary.txt
grep -E "xupd_tile|flux_kernel|flux_tile" $OUT/pmc_summary.txt | cut -c1-150
