#!/usr/bin/env python3
"""print the essentials of bench.py JSON lines: tools/show_bench.py gpurun_out/b_*.json"""
import json, sys
for f in sys.argv[1:]:
    try:
        b = json.loads([l for l in open(f) if l.startswith('{"metric"')][-1])     # a bench.py --detail file (or an old full line)
    except Exception as e:
        print(f, "unreadable:", e); continue
    c = b["config"]
    print("%s: %.3f G cell-updates/s, %.2f ms/step, nens %s nt %s, rows flagged %s/%s" % (
        f, b["value"] / 1e9, b["ms_per_step"], c["nens_per_gpu"], c["num_tracers"], c.get("fct_rows_flagged_last_stage"), c.get("fct_rows")))
    tot = 0.0
    for k in b.get("kernel_rooflines") or []:
        tot += k["ms_per_stage"]
        print("    %-22s %.3f ms/stage  own %.2f GB -> %.0f GB/s  valu_frac %.2f" % (k["kernel"], k["ms_per_stage"], k.get("own_bytes_per_stage", k.get("own_bytes_per_launch", 0)) / 1e9, k["own_GBps"], k["valu_frac"]))
    if tot:
        print("    stage total %.3f ms" % tot)
    for n, o in (b.get("other_configs") or {}).items():
        print("    other %s: %s" % (n, "%.3f G" % (o["value"] / 1e9) if o.get("value") else o))
