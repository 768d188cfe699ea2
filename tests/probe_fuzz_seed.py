"""One seed of tests/test_fuzz_parity.py taken apart (TEST INFRASTRUCTURE: uses the oracle and the host emulation): where the HIP path
and the emulation of its arithmetic differ from the oracle, field by field, and in which cells.
    python tests/probe_fuzz_seed.py SEED [lanes xkernels]"""
import copy
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_fuzz_parity as T  # noqa: E402
from emu_harness import EmuDycore  # noqa: E402
from oracle import awfl_oracle as ao  # noqa: E402
from pam_amd import idealized as idz  # noqa: E402


def main():
    import torch
    from pam_amd import Dycore, PamCoupler
    seed = int(sys.argv[1])
    c = T.draw_case(seed)
    if os.environ.get("PROBE_MODE"):
        c["mode_a"] = os.environ["PROBE_MODE"] == "A"
    if os.environ.get("PROBE_STEPS"):
        c["nsteps"] = int(os.environ["PROBE_STEPS"])
    print(T.describe(c))
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
    fo = copy.deepcopy(f)
    o = ao.OracleDycore(nens, nx, ny, nz, xlen, ylen, dz, pos, mass, idwv, consts=consts)
    if not c["mode_a"]:
        o.set_grav_balance(False)
    o.declare_current_profile_as_hydrostatic(fo)
    for _ in range(c["nsteps"]):
        o.time_step(fo, c["crm_dt"])
    fe = copy.deepcopy(f)
    g = EmuDycore(nens, nx, ny, nz, xlen, ylen, dz, pos, mass, idwv, consts=consts)
    if not c["mode_a"]:
        g.set_grav_balance(False)
    g.declare_current_profile_as_hydrostatic(fe)
    for _ in range(c["nsteps"]):
        g.time_step(fe, c["crm_dt"])
    coupler = PamCoupler("cuda:0")
    coupler.set_option("crm_dt", c["crm_dt"])
    for k, v in consts.items():
        coupler.set_option(k, v)
    coupler.allocate_coupler_state(nz, ny, nx, nens)
    coupler.set_grid(xlen, ylen, zi)
    for n, p, m in tr:
        coupler.add_tracer(n, "", p, m)
    d = Dycore()
    d.init(coupler)
    if len(sys.argv) > 3:
        d.set_lane_mapping(sys.argv[2], sys.argv[3])
    coupler.load_fields(copy.deepcopy(f))
    if not c["mode_a"]:
        coupler.set_option("balance_hydrostasis_with_gravity", False)
    d.declare_current_profile_as_hydrostatic(coupler)
    for _ in range(c["nsteps"]):
        d.timeStep(coupler)
    torch.cuda.synchronize()
    got = coupler.dump_fields()
    print("lane mapping", d.get_lane_mapping())
    for k in ("density_dry", "temp", "uvel", "vvel", "wvel"):
        s = np.abs(fo[k]).max()
        eg, ee, ge = np.abs(got[k] - fo[k]) / s, np.abs(fe[k] - fo[k]) / s, np.abs(got[k] - fe[k]) / s
        print("%-12s max|oracle| %.3e   hip-oracle %.2e   emu-oracle %.2e   hip-emu %.2e" % (k, s, eg.max(), ee.max(), ge.max()))
    k = "wvel"
    e = np.abs(got[k] - fo[k])
    idx = np.argsort(e.ravel())[::-1][:8]
    for i in idx:
        kk, j, ii, en = np.unravel_index(i, e.shape)
        print("  w cell k=%d j=%d i=%d e=%d: oracle % .6e  hip-oracle % .2e  emu-oracle % .2e" %
              (kk, j, ii, en, fo[k][kk, j, ii, en], got[k][kk, j, ii, en] - fo[k][kk, j, ii, en], fe[k][kk, j, ii, en] - fo[k][kk, j, ii, en]))
    print("levels: max |hip-oracle| of w per level:", ["%.1e" % v for v in e.max(axis=(1, 2, 3))])


if __name__ == "__main__":
    main()
