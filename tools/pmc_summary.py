#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (mean counter value per dispatch).

usage: tools/pmc_summary.py <counter_collection.csv> [...]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide
coalesced streams (MI355X_MICROARCH.md "HBM"): the guide's correction (x2) applies to 16-B-per-lane loads; our kernels
load 8 B per lane (512 B per wavefront instruction), an access width the guide calls uncalibrated -- both the raw and
the x2 figure are printed.
"""
import csv
import sys
from collections import defaultdict


def short(name):
    for k in ("awfl_flux_kernel", "awfl_update_kernel<1>", "awfl_update_kernel<2>", "awfl_update_kernel<3>",
              "awfl_fct_kernel", "awfl_init_prim_kernel", "awfl_finalize_kernel", "awfl_cfl_kernel", "awfl_hydro_kernel",
              "awfl_stage_kernel", "awfl_update_kernel_kt"):
        if k in name:
            return k
    return None


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for path in [a for a in sys.argv[1:] if not a.startswith('--')]:
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if k:
                    acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        for c in sorted(acc[k]):
            v = acc[k][c]
            mean = sum(v) / len(v)
            extra = ""
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                extra = "  = %.3f GB/launch" % (mean * 1024 / 1e9)
                if c == "FETCH_SIZE":
                    extra += "  (x2 gfx950 correction: %.3f GB)" % (2 * mean * 1024 / 1e9)
            print("%-26s %-22s n=%4d mean=%.6g%s" % (k, c, len(v), mean, extra))
    for k in sorted(acc):
        a = acc[k]
        if "GRBM_GUI_ACTIVE" in a and "SQ_INSTS_VALU" in a:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; a 64-lane fp64 VALU instruction occupies its SIMD for 4 cycles;
            # 256 CUs x 4 SIMDs
            cyc = sum(a["GRBM_GUI_ACTIVE"]) / len(a["GRBM_GUI_ACTIVE"]) / 8.0
            inst = sum(a["SQ_INSTS_VALU"]) / len(a["SQ_INSTS_VALU"])
            print("%-26s derived: %.3g busy cycles per XCD, VALU issue occupancy (4 cycles per wave-instruction, 1024 SIMDs) = %.1f %%"
                  % (k, cyc, 100.0 * inst * 4.0 / (cyc * 1024.0)))
    if "--flux-json" in sys.argv and "awfl_flux_kernel" in acc:
        import json
        f = acc["awfl_flux_kernel"]
        fetch = sum(f["FETCH_SIZE"]) / len(f["FETCH_SIZE"]) * 1024 * 2      # gfx950: x2 (calibrated)
        write = sum(f["WRITE_SIZE"]) / len(f["WRITE_SIZE"]) * 1024
        print(json.dumps({"hbm_bytes_per_launch": fetch + write, "fetch_bytes_x2_corrected": fetch, "write_bytes": write,
                          "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --chunks 1, config c2"}))


if __name__ == "__main__":
    main()
