#!/usr/bin/env python3
"""Table of a tools/bench_small.sh run: config, members, cell-updates/s, ms per step, stage kernels (ms per stage)."""
import json
import sys

for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("=="):
        print(line)
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    c = d["config"]
    ks = d.get("kernel_rooflines") or []
    kstr = " ".join("%s=%.4f" % (k["kernel"].replace("awfl_", "").replace("_kernel", ""), k["ms_per_stage"]) for k in ks)
    rf = d.get("roofline") or {}
    print("%-60s nens=%-5d %8.4f G  %9.3f ms/step  stage %.4f ms | %s" % (c["workload"][:60], c["nens_per_gpu"], d["value"] / 1e9,
                                                                     d["ms_per_step"], rf.get("stage_ms_back_to_back", 0.0), kstr))
