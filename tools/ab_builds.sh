#!/bin/bash
# A/B of two BUILDS of the library on one GPU box (run through gpurun from the repo root): put the two shared objects at
# pam_amd/libpam_amd_awfl_base.so and pam_amd/libpam_amd_awfl_new.so (built here: `*.so` is git-ignored but travels with the push), e.g.
#   cp pam_amd/libpam_amd_awfl.so pam_amd/libpam_amd_awfl_base.so; <edit a kernel>; python __graft_entry__.py; cp ... _new.so
# The script alternates them under `bench.py <args>` (default: C2, 5 steps) and prints the value and the stage kernels' ms per stage.
# Remember to restore pam_amd/libpam_amd_awfl.so (python __graft_entry__.py --force) afterwards.
ARGS=${@:-"--steps 5 --warmup 2"}
set -e
for r in 1 2 3; do
  for v in base new; do
    cp pam_amd/libpam_amd_awfl_$v.so pam_amd/libpam_amd_awfl.so
    python bench.py $ARGS --no-cpu-baseline --no-other-configs --detail gpurun_out/abso_$v.json > /dev/null 2>&1
    python - <<PY
import json
d=json.load(open("gpurun_out/abso_$v.json"))
print("round $r %-5s %.4f G  %s" % ("$v", d["value"]/1e9, " ".join("%s=%.3f"%(k["kernel"].replace("awfl_","").replace("_kernel",""),k["ms_per_stage"]) for k in d["kernel_rooflines"])))
PY
  done
done
