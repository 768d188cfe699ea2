#!/bin/bash
# Run the plain bench under the kernel tracer until the per-member C4 shard (or C2 at 128 members) shows its slow mode; keep that run's trace.
# usage (through gpurun, from the repo root): bash tools/probe/catch_slow_mode.sh [max runs]
R=$PWD; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 ${1:-6}); do
  rm -rf /tmp/kt_slow
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_slow -o t -- python3 $R/bench.py --detail /tmp/slow_detail.json > /tmp/slow_line.json 2>/dev/null || { echo "run $i failed"; exit 1; }
  python3 - <<PY
import json
d=json.load(open("/tmp/slow_line.json")); o=d["other"]
print("run $i: c4 %.4f c4_perens %.4f c2_shard128 %.4f" % (o["c4"]/1e9, o["c4_perens"]/1e9, o["c2_shard128"]/1e9), flush=True)
open("/tmp/slow_flag","w").write("1" if (o["c4_perens"] < 0.75e9 or o["c2_shard128"] < 2.3e9) else "0")
PY
  if [ "$(cat /tmp/slow_flag)" = "1" ]; then cp /tmp/kt_slow/t_kernel_trace.csv $R/gpurun_out/slow_mode_kernel_trace.csv; cp /tmp/slow_line.json $R/gpurun_out/slow_mode_line.json; echo "caught in run $i"; exit 0; fi
  cp /tmp/kt_slow/t_kernel_trace.csv $R/gpurun_out/fast_mode_kernel_trace.csv
done
echo "not caught"
