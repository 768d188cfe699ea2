"""Randomised parity sweep: the HIP path against the CPU oracle on SEEDED random shapes and options -- member counts on both sides of
every lane-mapping rule (1 .. 130: tile kernels with shuffles or LDS, flat lanes, ragged member lanes), line lengths from the minimum
(3 cells) upwards, 2-D and 3-D, uniform / stretched / per-member vertical grids, 1 .. 10 tracers with random `positive` / `adds_mass`
flags and water vapour at a random position, balance modes A and B, vapour limited or not, and (a third of the cases) a forced lane
mapping instead of the automatic one.  Every case goes through the repository's one gate (tests/parity_gate.py): rho_d, T and water
vapour at 1e-12 without exception; the noise-dominated fields on the measured curve 1e-11 (1 + nsub/3) -- and where a random shape
leaves that curve (seen: v = 0.06 m/s of pure noise on a 17x5x14 grid, 1.04 of the curve), the case must stay within 4x the ORACLE'S
OWN response to one ulp of input noise in T (two perturbed twin runs of the oracle), which is what the curve stands for; such seeds are
listed as "noise floor" in the sweep.

The hand-picked cases of test_gpu_parity.py name the paths they cover; this file is there for the combinations nobody thought of.
PAM_AMD_FUZZ_SEEDS=N runs N seeds instead of the default 16 (round 5: 400 seeds, profiles/r05_fuzz_parity.txt)."""
import copy
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))     # (for `python tests/test_fuzz_parity.py`)
from pam_amd import idealized as idz   # noqa: E402

pytestmark = pytest.mark.gpu

from parity_gate import TOL_TIGHT, tol_noise_fields, worst_errors   # noqa: E402

NSEEDS = int(os.environ.get("PAM_AMD_FUZZ_SEEDS", "16"))
MAX_CELLS = 24000          # the oracle runs ~2e5 cell-updates/s on one core


def draw_case(seed):
    """the case of one seed (pure function of the seed: a failing seed is a reproducible case)"""
    rng = np.random.default_rng(1000003 * seed + 17)
    while True:
        nens = int(rng.choice([1, 1, 2, 3, 4, 5, 8, 13, 16, 31, 32, 63, 64, 65, 70, 96, 130]))
        nx = int(rng.integers(3, 25))
        ny = int(rng.choice([1, 1, 1, 3, 4, 5, 7, 9]))
        nz = int(rng.integers(4, 15))
        if nens * nx * ny * nz <= MAX_CELLS:
            break
    nt = int(rng.choice([1, 1, 2, 3, 4, 5, 7, 10]))
    tr = [("q%02d" % i, bool(rng.random() < 0.7), bool(rng.random() < 0.5)) for i in range(nt - 1)]
    tr.insert(int(rng.integers(0, nt)), ("water_vapor", True, True))
    ztop = float(rng.choice([9000.0, 11000.0, 15000.0]))       # (not 12 km: the sounding's tropopause -- a level exactly there is 0/0)
    grid = str(rng.choice(["uniform", "stretched", "stretched"]))
    zint = idz.uniform_interfaces(nz, ztop) if grid == "uniform" else idz.stretched_interfaces(nz, ztop)
    c = dict(seed=seed, nens=nens, nx=nx, ny=ny, nz=nz, tracers=tr, zint=zint, grid=grid,
             per_ens=bool(rng.random() < 0.3), mode_a=bool(rng.random() < 0.6), dry_air=bool(rng.random() < 0.4),
             consts=idz.CONSTS_P3 if rng.random() < 0.3 else idz.CONSTS_DEFAULT,
             crm_dt=float(rng.choice([1.0, 2.0, 3.0])), nsteps=int(rng.integers(1, 3)),
             dxy=float(rng.choice([250.0, 500.0, 1000.0])),
             lanes=str(rng.choice(["auto", "auto", "auto", "auto", "member", "flat"])),
             xkernels=str(rng.choice(["auto", "auto", "auto", "auto", "sweep", "tile"])),
             xexchange=str(rng.choice(["auto", "auto", "lds"])),
             fusion=str(rng.choice(["auto", "auto", "separate", "inside", "beside"])))
    return c


def describe(c):
    return ("seed %d: nens %d, %dx%dx%d, nt %d (vapour at %d), %s%s, mode %s, dry_air %d, crm_dt %g x%d, dxy %g, lanes %s/%s/%s/%s"
            % (c["seed"], c["nens"], c["nx"], c["ny"], c["nz"], len(c["tracers"]), [t[0] for t in c["tracers"]].index("water_vapor"),
               c["grid"], "+per-member" if c["per_ens"] else "", "A" if c["mode_a"] else "B", c["dry_air"], c["crm_dt"], c["nsteps"],
               c["dxy"], c["lanes"], c["xkernels"], c["xexchange"], c["fusion"]))


class IllPosedCase(Exception):
    """the drawn case is outside what the reference algorithm itself can run"""


def run_case(c):
    import torch
    from pam_amd import Dycore, PamCoupler
    from pam_amd.capi import PamAmdError
    from oracle import awfl_oracle as ao
    nens, nx, ny, nz, tr, consts = c["nens"], c["nx"], c["ny"], c["nz"], c["tracers"], c["consts"]
    names, pos, mass, idwv = idz.tracer_flags(tr)
    xlen = nx * c["dxy"]
    ylen = ny * c["dxy"] if ny > 1 else xlen
    f = idz.supercell_fields(nens, nx, ny, nz, c["zint"], consts=consts, tracers=tr, magnitude=0.5, id0=c["seed"])
    idz.add_tracer_blobs(f, tr, xlen, ylen, c["zint"])
    if c["dry_air"]:
        f["uvel"] -= 25.0
        f["vvel"] += 7.0 if ny > 1 else 0.0
        idz.carve_dry_air(f, tr)
    zi = np.asarray(c["zint"])[:, None] * np.ones((1, nens))
    if c["per_ens"]:
        zi = zi * (1 + 0.01 * np.arange(nens))[None, :]
    dz = np.diff(zi, axis=0)
    coupler = PamCoupler("cuda:0")
    coupler.set_option("crm_dt", c["crm_dt"])
    for k, v in consts.items():
        coupler.set_option(k, v)
    coupler.allocate_coupler_state(nz, ny, nx, nens)
    coupler.set_grid(xlen, ylen, zi)
    for n, p, m in tr:
        coupler.add_tracer(n, "", p, m)
    dycore = Dycore()
    dycore.init(coupler)
    applied = []
    for what, forced, call in (("lanes", (c["lanes"], c["xkernels"]) != ("auto", "auto"), lambda: dycore.set_lane_mapping(c["lanes"], c["xkernels"])),
                               ("xexchange", c["xexchange"] != "auto", lambda: dycore.set_x_exchange(c["xexchange"])),
                               ("fusion", c["fusion"] != "auto", lambda: dycore.set_tile_fusion(c["fusion"]))):
        if not forced:
            continue
        try:                               # (a mapping the shape does not support is refused with an error: the case then runs automatic)
            call()
            applied.append(what)
        except PamAmdError:
            applied.append(what + " refused")
    coupler.load_fields(f)
    fo = copy.deepcopy(f)

    def oracle_run(fields, beside=None):
        oracle = ao.OracleDycore(nens, nx, ny, nz, xlen, ylen, dz, pos, mass, idwv, consts=consts)
        if not c["mode_a"]:
            oracle.set_grav_balance(False)
        oracle.declare_current_profile_as_hydrostatic(fields)
        n = 0
        for _ in range(c["nsteps"]):
            n_gpu = beside() if beside else None
            n_cpu, dt_cpu = oracle.time_step(fields, c["crm_dt"])
            if beside:
                assert n_gpu == n_cpu, (n_gpu, n_cpu)
                assert abs(dycore.last_dt_dyn - dt_cpu) <= 1e-15 * dt_cpu
            n += n_cpu
        return n

    if c["per_ens"] and nens > 60:
        # members 60+ of a per-member case sit on a grid 1.6 ... 2.3 x the base grid: with 4-5 levels over 15 km the reference's own
        # arithmetic leaves the numbers there (top layers of > 7 km: the extrapolated ghost pressure of declare_current_profile_as_
        # hydrostatic is not positive -- NaN in variable_gravity; seeds 2634, 3494, 3980).  Such a case says nothing about parity.
        probe = copy.deepcopy(f)
        oracle_run(probe)
        if not all(np.isfinite(probe[k]).all() for k in ("density_dry", "uvel", "vvel", "wvel", "temp")):
            dycore.finalize(coupler)
            raise IllPosedCase("the oracle itself is not finite on this grid")
    if not c["mode_a"]:
        coupler.set_option("balance_hydrostasis_with_gravity", False)
    dycore.declare_current_profile_as_hydrostatic(coupler)
    nsub = oracle_run(fo, beside=lambda: dycore.timeStep(coupler))
    torch.cuda.synchronize()
    got = coupler.dump_fields()
    mapping = dycore.get_lane_mapping()
    dycore.finalize(coupler)
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp"):
        assert np.isfinite(fo[k]).all() and np.isfinite(got[k]).all(), k
    # (grids of 3-4 cells per direction wrap the periodic stencil over the whole line: their noise sits higher on the measured curve,
    # like test_gpu_parity's `3d_minimal_1x3x3x3`)
    factor = 4.0 if min(nx, nz, ny if ny > 1 else nx) <= 4 else 1.0
    worst = worst_errors(got, fo, names)
    loose = tol_noise_fields(nsub, factor)
    tight = lambda k: k.split("_elementwise")[0] in ("density_dry", "temp", "water_vapor")      # noqa: E731
    for k, e in worst.items():
        if tight(k):
            assert e <= TOL_TIGHT, (k, e, TOL_TIGHT, worst)
    over = [k for k, e in worst.items() if not tight(k) and not e <= loose]
    floor = {}
    if over:          # off the measured curve: the flow's own sensitivity decides (the oracle against itself, 1 ulp of noise in T)
        rng = np.random.default_rng(c["seed"])
        for _ in range(2):
            twin = copy.deepcopy(f)
            twin["temp"] = twin["temp"] * (1.0 + rng.integers(-1, 2, size=twin["temp"].shape) * 1.1e-16)
            oracle_run(twin)
            for k, e in worst_errors(twin, fo, names).items():
                floor[k] = max(floor.get(k, 0.0), e)
        for k in over:
            assert worst[k] <= 4.0 * floor[k], (k, worst[k], "gate", loose, "4 x oracle's own noise response", 4.0 * floor[k], worst)
    return nsub, mapping, applied, {k: (worst[k], loose, floor[k]) for k in over}


@pytest.mark.parametrize("seed", range(NSEEDS))
def test_random_case_matches_oracle(seed):
    c = draw_case(seed)
    try:
        run_case(c)
    except IllPosedCase as e:
        pytest.skip(describe(c) + ": " + str(e))
    except AssertionError as e:
        raise AssertionError(describe(c) + "\n" + str(e)) from e


if __name__ == "__main__":      # python tests/test_fuzz_parity.py FIRST COUNT: a sweep with one line per seed (tests/ may use the oracle)
    first, count = int(sys.argv[1]), int(sys.argv[2])
    bad = skipped = 0
    for s in range(first, first + count):
        c = draw_case(s)
        try:
            try:
                nsub, mapping, applied, over = run_case(c)
            except IllPosedCase as e:
                skipped += 1
                print("skip %s | %s" % (describe(c), e), flush=True)
                continue
            note = "".join("; noise floor: %s %.2e (curve %.1e, oracle's own response to 1 ulp %.2e)" % ((k,) + v) for k, v in over.items())
            print("ok   %s | %d sub-steps, forced %s%s" % (describe(c), nsub, ",".join(applied) or "-", note), flush=True)
        except Exception as e:        # noqa: BLE001  (a sweep reports every seed)
            bad += 1
            print("FAIL %s | %s" % (describe(c), str(e).splitlines()[0][:300]), flush=True)
    print("%d seeds, %d failed%s" % (count, bad, (", %d skipped (the oracle itself not finite)" % skipped) if skipped else ""))
    sys.exit(1 if bad else 0)
