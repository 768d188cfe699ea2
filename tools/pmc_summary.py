#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (mean counter value per dispatch).

usage: tools/pmc_summary.py <counter_collection.csv> [...]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide
coalesced streams (MI355X_MICROARCH.md "HBM"): the guide's correction (x2) applies to 16-B-per-lane loads; our kernels
load 8 B per lane (512 B per wavefront instruction), an access width the guide calls uncalibrated -- both the raw and
the x2 figure are printed.
"""
import csv
import os
import sys
from collections import defaultdict


def short(name):
    """awfl_flux_kernel<false, true>(...) -> awfl_flux_kernel<false, true>; other kernels (torch fills etc.) are skipped"""
    import re
    m = re.search(r"(awfl_\w+_kernel(?:<[^>]*>)?)", name)
    return m.group(1) if m else None


def main():
    acc = defaultdict(lambda: defaultdict(list))
    args = sys.argv[1:]
    if "--traffic-json" in args:   # the flag's value (content hash of pam_amd/csrc) is not a file
        i = args.index("--traffic-json")
        args = args[:i] + args[i + 2:]
    for path in [a for a in args if not a.startswith('--')]:
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
            print("%-34s %-22s n=%4d mean=%.6g%s" % (k, c, len(v), mean, extra))
    for k in sorted(acc):
        a = acc[k]
        if "GRBM_GUI_ACTIVE" in a and "SQ_INSTS_VALU" in a:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; a 64-lane fp64 VALU instruction occupies its SIMD for 4 cycles;
            # 256 CUs x 4 SIMDs
            cyc = sum(a["GRBM_GUI_ACTIVE"]) / len(a["GRBM_GUI_ACTIVE"]) / 8.0
            inst = sum(a["SQ_INSTS_VALU"]) / len(a["SQ_INSTS_VALU"])
            print("%-34s derived: %.3g busy cycles per XCD, VALU issue occupancy (4 cycles per wave-instruction, 1024 SIMDs) = %.1f %%"
                  % (k, cyc, 100.0 * inst * 4.0 / (cyc * 1024.0)))
    for k in sorted(acc):
        a = acc[k]
        if "SQ_WAVE_CYCLES" in a and "SQ_WAIT_ANY" in a:
            m = {c: sum(v) / len(v) for c, v in a.items()}
            wc = m["SQ_WAVE_CYCLES"]
            print("%-34s wave-cycle split: waiting (s_waitcnt/barrier) %.1f %%, issue-stalled %.1f %%, issuing %.1f %%"
                  % (k, 100 * m["SQ_WAIT_ANY"] / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc))
    for k in sorted(acc):
        a = acc[k]
        if "SQ_WAVES" in a and "SQ_WAVE_CYCLES" in a:
            # per wavefront: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md, cycle constants)
            m = {c: sum(v) / len(v) for c, v in a.items()}
            w = max(m["SQ_WAVES"], 1.0)
            extra = ""
            for c, label in (("SQ_INSTS_LDS", "LDS"), ("SQ_INSTS_SALU", "SALU"), ("SQ_INSTS_SMEM", "SMEM"), ("SQ_INSTS_VMEM_RD", "vmem-rd"),
                             ("SQ_INSTS_VMEM_WR", "vmem-wr"), ("SQ_IFETCH", "ifetch")):
                if c in m:
                    extra += ", %s %.0f" % (label, m[c] / w)
            print("%-34s per wavefront: %.0f wavefronts per launch, %.0f cycles resident, VALU instructions %.0f%s"
                  % (k, w, 4.0 * m["SQ_WAVE_CYCLES"] / w, m.get("SQ_INSTS_VALU", 0.0) / w, extra))
    if "--traffic-json" in sys.argv:
        # HBM bytes per launch of every stage kernel: FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B; calibrated on
        # kernels of known byte counts) + WRITE_SIZE; template instances of one kernel are averaged
        import json, re

        def family(k):
            """template instances of one kernel are one family -- except the two PHASES of the tracer sweeps, which are different
            kernels in all but name: awfl_xtr_kernel<STAGE, PHASE> -> awfl_xtr_kernel<PHASE> (bench.py's name for them)"""
            m = re.match(r"(awfl_xtrn?(?:_tile)?_kernel)<\s*\d+\s*,\s*(\d+)\s*[,>]", k)      # (further template arguments: variants of the phase)
            return "%s<%s>" % (m.group(1), m.group(2)) if m else re.sub(r"<.*", "", k)
        out = {}
        for k in acc:
            if "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
                base = family(k)
                f, w = acc[k]["FETCH_SIZE"], acc[k]["WRITE_SIZE"]
                out.setdefault(base, []).append((sum(f) * 1024 * 2, sum(w) * 1024, len(f)))
        res = {}
        nlaunch = {base: sum(r[2] for r in rows) for base, rows in out.items()}
        # tendency stages in the profiled run: every stage has exactly one update-type launch
        nstage = nlaunch.get("awfl_xupd_kernel") or nlaunch.get("awfl_xupd_tile_kernel") or nlaunch.get("awfl_update_kernel") or 1
        for base, rows in out.items():
            n = nlaunch[base]
            fetch, write = sum(r[0] for r in rows), sum(r[1] for r in rows)
            res[base] = {"hbm_bytes_per_launch": (fetch + write) / n, "launches": n, "launches_per_stage": n / nstage,
                         "hbm_bytes_per_stage": (fetch + write) / nstage, "fetch_bytes_x2_corrected_per_stage": fetch / nstage,
                         "write_bytes_per_stage": write / nstage}
            # VALU issue: wave-level instructions and busy cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs), all template instances
            inst = sum(sum(acc[k]["SQ_INSTS_VALU"]) for k in acc if family(k) == base and "SQ_INSTS_VALU" in acc[k])
            busy = sum(sum(acc[k]["GRBM_GUI_ACTIVE"]) for k in acc if family(k) == base and "GRBM_GUI_ACTIVE" in acc[k])
            if inst and busy:
                res[base]["valu_insts_per_stage"] = inst / nstage
                res[base]["busy_cycles_per_xcd_per_stage"] = busy / 8.0 / nstage
        i = sys.argv.index("--traffic-json")
        print(json.dumps({"csrc_hash": sys.argv[i + 1], "compute_units": int(os.environ.get("PMC_COMPUTE_UNITS", "0")) or None, "kernels": res,
                          "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py, config %s" % os.environ.get("PMC_SOURCE_CONFIG", "c2")}))


if __name__ == "__main__":
    main()
