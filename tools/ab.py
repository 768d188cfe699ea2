#!/usr/bin/env python3
"""A/B (/C ...) timing of bench.py variants on ONE box, alternating the variants so that box-to-box and drift effects cancel.

  tools/ab.py [--rounds R] [--out FILE] --common "<bench.py args shared by all>" --variant name="<extra args>" [--variant ...]

Every run is `python bench.py <common> <extra> --no-cpu-baseline --no-other-configs`; printed per variant: cell-updates/s of every round,
the median, and the stage kernels' ms per stage of the last round (HIP events; omitted with --no-kernel-timing in <common>).
Replaces the one-off tools/exp_*.sh scripts of rounds 2-4 (each was this loop with the variants written out)."""
import argparse
import json
import os
import shlex
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--common", default="")
    ap.add_argument("--variant", action="append", default=[], help='name="extra bench.py arguments"')
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    variants = []
    for v in a.variant:
        name, _, extra = v.partition("=")
        variants.append((name, extra))
    if not variants:
        sys.exit("tools/ab.py: no --variant")
    res = {n: [] for n, _ in variants}
    last = {}
    detail = os.path.join(ROOT, "gpurun_out", "ab_detail.json")
    os.makedirs(os.path.dirname(detail), exist_ok=True)
    for r in range(a.rounds):
        for name, extra in variants:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + shlex.split(a.common) + shlex.split(extra) + [
                "--no-cpu-baseline", "--no-other-configs", "--detail", detail]
            p = subprocess.run(cmd, capture_output=True, text=True)
            if p.returncode:
                print("%s: FAILED: %s" % (name, p.stderr[-500:]))
                res[name].append(float("nan"))
                continue
            d = json.load(open(detail))
            res[name].append(d["value"] / 1e9)
            last[name] = d
            print("round %d %-28s %.4f G  %.3f ms/step" % (r, name, d["value"] / 1e9, d["ms_per_step"]), flush=True)
    lines = []
    for name, _ in variants:
        v = [x for x in res[name] if x == x]
        ks = " ".join("%s=%.4f" % (k["kernel"].replace("awfl_", "").replace("_kernel", ""), k["ms_per_stage"])
                      for k in (last.get(name, {}).get("kernel_rooflines") or []))
        kk = last.get(name, {}).get("kernels") or {}
        nst = max(1, (kk.get("xupd") or kk.get("update") or {}).get("launches", 1))
        for extra in ("flux_xy", "flux_z"):      # the y and z sweeps as launches of their own (folded stage / per-member grids): shipped schedule
            if extra in kk:
                ks += " %s=%.4f" % (extra, kk[extra]["total_ms"] / nst)
        st = (last.get(name, {}).get("roofline") or {}).get("stage_ms_back_to_back")
        lines.append("%-28s median %.4f G  [%s]  stage %s ms | %s" % (name, statistics.median(v) if v else float("nan"),
                                                                     " ".join("%.4f" % x for x in res[name]),
                                                                     ("%.4f" % st) if st else "-", ks))
    print("\n".join(lines))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "a") as fh:
            fh.write("# common: %s\n" % a.common)
            fh.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
