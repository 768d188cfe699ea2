#!/bin/bash
# ONE regeneration of the round's rocprofv3 evidence (run through gpurun from the repo root, after the last kernel change):
#   tools/collect_profiles.sh r06      -> gpurun_out/profiles_r06/ : copy its files into profiles/ (tracked)
# Workloads: C2 (one member range per launch: the launches `roofline` is computed from), C2 on the default schedule (kernel trace only),
# C3, C4 (one GPU's 512-member shard), and the small-ensemble workloads whose x direction runs as tile kernels: the C2 grid with 1, 8 and
# 32 members and the reference's own input shape (250 x 1 x 50, nens = 1, NT = 4).  Per workload: kernel_stats.csv, the bench line and
# detail object printed under the profiler, the PMC passes (FETCH_SIZE, WRITE_SIZE, two SQ sets, GRBM_GUI_ACTIVE), pmc_summary.txt and
# traffic.json (keyed by the content hash of pam_amd/csrc: bench.py reports `traffic` only for the build it was measured on).
set -e
round=${1:-r06}
R=$PWD
DST=$R/gpurun_out/profiles_$round
# PROFILE_SET=a (the BASELINE configurations + per-member grids) | b (small ensembles + the default schedule) | all: two gpurun calls of
# at most 20 minutes each cover the whole set
SET=${PROFILE_SET:-all}
[ "$SET" != b ] && rm -rf $DST
mkdir -p $DST
take() {   # take <tag> <prefix>: move the files of one tools/profile.sh run under their tracked names
  local src=$R/gpurun_out/prof_$1 p=$2
  cp $src/kernel_stats.csv $DST/${round}_${p}_kernel_stats.csv
  cp $src/bench_under_rocprof.json $DST/${round}_${p}_bench_under_rocprof.json
  cp $src/pmc_summary.txt $DST/${round}_${p}_pmc_summary.txt
  cp $src/traffic.json $DST/${round}_${p}_traffic.json
  for f in $src/pmc_*.csv; do
    # the raw counter tables are large (one row per dispatch and counter): keep them for the headline config only
    [ "$p" = c2 ] && cp $f $DST/${round}_${p}_$(basename $f)
  done
  true
}
if [ "$SET" != b ]; then
tools/profile.sh c2 --chunks 1;                                   take c2 c2
tools/profile.sh c3 --quick --config c3 --chunks 1;               take c3 c3
tools/profile.sh c4 --quick --config c4 --chunks 1;               take c4 c4
# per-member vertical grids (awfl_fluxz_pe_kernel): C2 and one GPU's shard of C4
tools/profile.sh c2_perens --quick --perens 1 --chunks 1;                take c2_perens c2_perens
tools/profile.sh c4_perens --quick --config c4 --perens 1 --chunks 1;    take c4_perens c4_perens
fi
if [ "$SET" != a ]; then
for n in 1 8 32; do
  tools/profile.sh c2grid_nens$n --config c2 --nens $n --steps 20 --warmup 2;   take c2grid_nens$n c2grid_nens$n
done
tools/profile.sh ref_nens1 --config ref --steps 20 --warmup 2;    take ref_nens1 ref_nens1
# the default schedule of the plain bench command (two independent member ranges): kernel trace only
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks_default
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_default -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --detail $DST/unused.json > $DST/default_schedule.log 2>&1 || { echo "default-schedule pass failed:"; tail -20 $DST/default_schedule.log; exit 1; }
rm -f $DST/default_schedule.log
cp $(find /tmp/ks_default -name '*kernel_stats.csv') $DST/${round}_c2_kernel_stats_default_schedule.csv
rm -f $DST/unused.json
cd $R
fi
ls $DST
