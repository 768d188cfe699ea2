"""Randomised schedule sweep: every launch-shape / lane-mapping / schedule knob of the C ABI is a pure scheduling device -- whatever
combination is selected, a timeStep must give the SAME BITS as member lanes + sweep kernels on one member range (the arithmetic of a
cell is the same code on the same values; tests/test_lane_mapping.py and tests/test_sharding_gpu.py check hand-picked combinations).
A seed draws a shape (1 .. 200 members, lines of 3 .. 48 cells, 2-D / 3-D, 1 .. 10 tracers, both balance modes, vapour limited or
not) and a handful of random knob settings: y/z lanes, x kernels, x tile geometry, exchange by LDS or shuffles, the flux tile kernel
and its tile sizes, parts beside / behind each other, the tile fusions, graph replay, member ranges and their schedule, flux segment
and span, tracers per wavefront, launch-tuning thresholds, fused / three-kernel stage.  A setting the shape does not support must be
REFUSED with an error (the run then keeps the previous setting); anything accepted must reproduce the reference run bit for bit over
two timeSteps of different length.  No oracle here: sizes are whatever the GPU does in a blink.

PAM_AMD_FUZZ_SEEDS=N runs N seeds instead of the default 12 (round 5: profiles/r05_fuzz_mappings.txt)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))     # (for `python tests/test_fuzz_mappings.py`)
from pam_amd import idealized as idz   # noqa: E402

pytestmark = pytest.mark.gpu

NSEEDS = int(os.environ.get("PAM_AMD_FUZZ_SEEDS", "12"))
NVARIANTS = 5
MAX_CELLS = 400000


def draw_shape(rng, seed):
    while True:
        nens = int(rng.choice([1, 1, 2, 3, 4, 7, 8, 16, 24, 32, 48, 63, 64, 65, 70, 128, 130, 192, 200]))
        nx = int(rng.integers(3, 49))
        ny = int(rng.choice([1, 1, 1, 3, 4, 6, 8, 10]))
        nz = int(rng.integers(4, 25))
        if nens * nx * ny * nz <= MAX_CELLS:
            break
    nt = int(rng.choice([1, 1, 2, 3, 4, 5, 7, 10]))
    tr = [("q%02d" % i, bool(rng.random() < 0.7), bool(rng.random() < 0.5)) for i in range(nt - 1)]
    tr.insert(int(rng.integers(0, nt)), ("water_vapor", True, True))
    ztop = float(rng.choice([9000.0, 11000.0, 15000.0]))
    grid = str(rng.choice(["uniform", "stretched", "stretched"]))
    return dict(seed=seed, nens=nens, nx=nx, ny=ny, nz=nz, tracers=tr, grid=grid,
                zint=idz.uniform_interfaces(nz, ztop) if grid == "uniform" else idz.stretched_interfaces(nz, ztop),
                per_ens=bool(rng.random() < 0.3), mode_a=bool(rng.random() < 0.6), dry_air=bool(rng.random() < 0.5),
                consts=idz.CONSTS_P3 if rng.random() < 0.3 else idz.CONSTS_DEFAULT, dxy=float(rng.choice([250.0, 500.0, 1000.0])))


def draw_knobs(rng, s):
    pick = lambda *v: v[int(rng.integers(0, len(v)))]      # noqa: E731
    k = {}
    if rng.random() < 0.8:
        k["lane_mapping"] = (pick("auto", "member", "flat"), pick("auto", "sweep", "tile", "tile"))
    if rng.random() < 0.4:
        k["x_tile"] = (pick(0, 0, 1, 2, 4, 8, 16, 64), pick(0, 0, 3, 4, 5, 8, s["nx"]), pick(0, 0, 1, 2, 4, 8))
    if rng.random() < 0.5:
        k["x_exchange"] = (pick("lds", "shuffle", "auto"),)
    if rng.random() < 0.6:
        k["flux_tile"] = (pick("auto", "sweep", "tile", "tile"), pick(0, 0, 2, 3, 5, 11), pick(0, 0, 2, 3, 6, 14))
    if rng.random() < 0.4:
        k["flux_tile_parts"] = (pick("behind", "beside"),)
    if rng.random() < 0.4:
        k["tile_state_parts"] = (pick("one", "parts"),)
    if rng.random() < 0.6:
        k["tile_fusion"] = (pick("separate", "inside", "beside"),)
    if rng.random() < 0.5:
        k["graph_replay"] = (pick("off", "on", "on"),)
    if rng.random() < 0.5:
        k["ensemble_chunks"] = (pick(0, 1, 2, 3, 4),)
    if rng.random() < 0.3:
        k["range_schedule"] = (pick(True, False),)
    if rng.random() < 0.4:
        k["flux_segment"] = (pick(1, 3, 8, 16, 64),)
    if rng.random() < 0.4:
        k["flux_span"] = (pick(0, 1, 4, 6, 16),)
    if rng.random() < 0.4:
        g = pick(0, 1, 2, 4)
        k["tracer_grouping"] = (g, bool(g == 2 and rng.random() < 0.5))
    if rng.random() < 0.4:
        k["launch_tuning"] = (pick(0, 64, 100000), pick(-1, 0, 10 ** 9), pick(-1, 0, 10 ** 9))
    if rng.random() < 0.15:
        k["fused_stage"] = (False,)
    if rng.random() < 0.4:        # (round 6) tracer phase 2 + pressure pass + fix-up as one launch or three
        k["tail_fusion"] = (pick("on", "off"),)
    if rng.random() < 0.4:        # (round 6) the y differences of the state folded into the z sweep's output: 3-D member-lane stages only
        k["yz_fold"] = (pick("on", "on", "off"),)
    return k


def draw_case(seed):
    rng = np.random.default_rng(7000003 * seed + 5)
    s = draw_shape(rng, seed)
    s["variants"] = [draw_knobs(rng, s) for _ in range(NVARIANTS)]
    return s


def describe(s):
    return ("seed %d: nens %d, %dx%dx%d, nt %d (vapour at %d), %s%s, mode %s, dry_air %d"
            % (s["seed"], s["nens"], s["nx"], s["ny"], s["nz"], len(s["tracers"]), [t[0] for t in s["tracers"]].index("water_vapor"),
               s["grid"], "+per-member" if s["per_ens"] else "", "A" if s["mode_a"] else "B", s["dry_air"]))


# (order matters where one knob's validity depends on another: the lane mapping first)
ORDER = ["fused_stage", "lane_mapping", "x_tile", "x_exchange", "flux_tile", "flux_tile_parts", "tile_state_parts", "tile_fusion",
         "graph_replay", "ensemble_chunks", "range_schedule", "flux_segment", "flux_span", "tracer_grouping", "launch_tuning", "tail_fusion", "yz_fold"]


def run(s, f, xlen, ylen, knobs):
    import torch
    from pam_amd import Dycore, PamCoupler
    from pam_amd.capi import PamAmdError
    nens, nx, ny, nz = s["nens"], s["nx"], s["ny"], s["nz"]
    zi = np.asarray(s["zint"])[:, None] * np.ones((1, nens))
    if s["per_ens"]:
        zi = zi * (1 + 0.01 * (np.arange(nens) % 16))[None, :]      # (the fields are one sounding: grids stretched further than this blow up)
    coupler = PamCoupler("cuda:0")
    coupler.set_option("crm_dt", 2.0)
    for k, v in s["consts"].items():
        coupler.set_option(k, v)
    coupler.allocate_coupler_state(nz, ny, nx, nens)
    coupler.set_grid(xlen, ylen, zi)
    for n, p, m in s["tracers"]:
        coupler.add_tracer(n, "", p, m)
    dycore = Dycore()
    dycore.init(coupler)
    refused = []
    for name in ORDER:
        if name in knobs:
            try:
                getattr(dycore, "set_" + name)(*knobs[name])
            except PamAmdError:
                refused.append(name)
    coupler.load_fields(f)
    if not s["mode_a"]:
        coupler.set_option("balance_hydrostasis_with_gravity", False)
    dycore.declare_current_profile_as_hydrostatic(coupler)
    ncyc = []
    for crm_dt in (2.0, 0.7):
        coupler.set_option("crm_dt", crm_dt)
        ncyc.append(dycore.timeStep(coupler))
    torch.cuda.synchronize()
    out = coupler.dump_fields()
    mapping = dycore.get_lane_mapping()
    dycore.finalize(coupler)
    return ncyc, out, mapping, refused


def run_case(s):
    nens, nx, ny, nz, tr = s["nens"], s["nx"], s["ny"], s["nz"], s["tracers"]
    xlen = nx * s["dxy"]
    ylen = ny * s["dxy"] if ny > 1 else xlen
    f = idz.supercell_fields(nens, nx, ny, nz, s["zint"], consts=s["consts"], tracers=tr, magnitude=0.5, id0=s["seed"])
    idz.add_tracer_blobs(f, tr, xlen, ylen, s["zint"])
    if s["dry_air"]:
        f["uvel"] -= 25.0
        f["vvel"] += 7.0 if ny > 1 else 0.0
        idz.carve_dry_air(f, tr)
    n0, ref, _, _ = run(s, f, xlen, ylen, {"lane_mapping": ("member", "sweep"), "ensemble_chunks": (1,), "graph_replay": ("off",)})
    for k in ("density_dry", "uvel", "wvel", "temp"):
        assert np.isfinite(ref[k]).all(), k
    notes = []
    for knobs in s["variants"]:
        n, out, mapping, refused = run(s, f, xlen, ylen, knobs)
        assert n == n0, (knobs, n, n0)
        for k in ("density_dry", "uvel", "vvel", "wvel", "temp"):
            assert np.array_equal(out[k], ref[k]), (k, knobs, mapping, refused)
        for t in range(len(tr)):
            assert np.array_equal(out["tracers"][t], ref["tracers"][t]), (tr[t][0], knobs, mapping, refused)
        notes.append("%s%s" % ("flat" if mapping["yz_flat"] else "member", "+tiles" if mapping["x_tiles"] else "+sweeps")
                     + ("(shuffle)" if mapping.get("x_shuffles") else "") + ("[refused: %s]" % ",".join(refused) if refused else ""))
    return notes


@pytest.mark.parametrize("seed", range(NSEEDS))
def test_random_schedules_give_the_same_bits(seed):
    s = draw_case(seed)
    try:
        run_case(s)
    except AssertionError as e:
        raise AssertionError(describe(s) + "\n" + str(e)[:2000]) from e


if __name__ == "__main__":      # python tests/test_fuzz_mappings.py FIRST COUNT: one line per seed
    first, count = int(sys.argv[1]), int(sys.argv[2])
    bad = 0
    for sd in range(first, first + count):
        s = draw_case(sd)
        try:
            notes = run_case(s)
            print("ok   %s | %s" % (describe(s), " ".join(notes)), flush=True)
        except Exception as e:        # noqa: BLE001
            bad += 1
            print("FAIL %s | %s" % (describe(s), str(e)[:600].replace("\n", " ")), flush=True)
    print("%d seeds x %d schedules, %d failed" % (count, NVARIANTS, bad))
    sys.exit(1 if bad else 0)
