"""Determinism / race check at the full sizes of C2 (with vapour limited: --limiter input), C3 and C4: the same N timeSteps run
from the same inputs -- with the default schedule (two independent member ranges), with one range, with four ranges on a shared
compute stream -- must give bit-identical coupler fields, and all must equal the three-kernel stage (which shares no kernel with the x-sweeps' read-backs
of their own stores, the two-phase tracer sweeps or the line-driven fix-up).  Round 6: a fifth schedule with the round's options the other
way round (the y differences folded into the z sweep on 3-D grids; the many-tracer tail as one launch / as three), and two configurations
with per-member vertical grids (awfl_fluxz_pe_kernel: tables staged in LDS, a barrier per level).   usage: tools/soak_configs.py [nsteps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pam_amd import Dycore, PamCoupler, idealized as idz

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
CONFIGS = {"c2_limiter": (1024, 32, 32, idz.TRACERS_NONE, idz.CONSTS_DEFAULT, True),
           "c3": (4096, 32, 1, idz.TRACERS_KESSLER_SHOC, idz.CONSTS_DEFAULT, False),
           "c4": (512, 32, 1, idz.TRACERS_P3_SHOC, idz.CONSTS_P3, False),
           # 3-D with many tracers (the y faces of the two-phase tracer sweeps) and vapour limited as well
           "3d_nt4_limiter": (256, 32, 32, idz.TRACERS_KESSLER_SHOC, idz.CONSTS_DEFAULT, True),
           "3d_nt10_p3": (128, 32, 32, idz.TRACERS_P3_SHOC, idz.CONSTS_P3, False),
           "c2_perens_limiter": (1024, 32, 32, idz.TRACERS_NONE, idz.CONSTS_DEFAULT, True),
           "c4_perens": (512, 32, 1, idz.TRACERS_P3_SHOC, idz.CONSTS_P3, False)}
allok = True
for name, (nens, nx, ny, tr, consts, dry) in CONFIGS.items():
    nz, zint = 60, idz.l60_interfaces()
    xlen = nx * 1000.0
    f = idz.supercell_fields(16, nx, ny, nz, zint, consts=consts, tracers=tr, magnitude=0.5)
    if len(tr) > 1:
        idz.add_tracer_blobs(f, tr, xlen, xlen, zint)
    if dry:
        idz.carve_dry_air(f, tr)
    c = PamCoupler("cuda:0"); c.set_option("crm_dt", 2.0)
    for k, v in consts.items(): c.set_option(k, v)
    c.allocate_coupler_state(nz, ny, nx, nens)
    if "perens" in name:      # every member on its own vertical grid (as bench.py --perens 1)
        a_ = 0.02 * (((np.arange(nens) * 37) % 101) - 50.0) / 50.0
        c.set_grid(xlen, xlen, np.asarray(zint)[:, None] * (1.0 + a_[None, :] * (1.0 - np.arange(nz + 1) / float(nz))[:, None]))
    else:
        c.set_grid(xlen, xlen, zint)
    for n, p, m in tr: c.add_tracer(n, "", p, m)
    d = Dycore(); d.init(c)
    names = ["density_dry", "uvel", "vvel", "wvel", "temp"] + c.get_tracer_names()
    init = {k: torch.from_numpy(f[k]).to("cuda:0").repeat(1, 1, 1, nens // 16).contiguous() for k in names[:5]}
    for t, n in enumerate(c.get_tracer_names()):
        init[n] = torch.from_numpy(f["tracers"][t]).to("cuda:0").repeat(1, 1, 1, nens // 16).contiguous()

    def run(fused, chunks):
        for k in names: c.dm.get(k).copy_(init[k])
        d.set_fused_stage(fused); d.set_ensemble_chunks(chunks)
        d.declare_current_profile_as_hydrostatic(c)
        tot = sum(d.timeStep(c) for _ in range(nsteps))
        torch.cuda.synchronize()
        return tot, {k: c.dm.get(k, readonly=True).clone() for k in names}, d.debug_fct_rows()

    na, a, rows = run(True, 0)          # default schedule: two independent member ranges
    nb, b, _ = run(True, 1)             # one range
    nc, cc, _ = run(False, 1)           # three-kernel stage
    d.set_range_schedule(False)
    nd, dd, _ = run(True, 4)            # four ranges, the polynomial kernels on one shared compute stream
    d.set_range_schedule(True)
    # round 6's options the other way round: fold on (3-D member lanes; refused elsewhere), tail fusion forced on or off
    try:
        d.set_yz_fold("on")
    except Exception:
        pass
    d.set_tail_fusion("off" if nens * nx * ny <= 512 * 32 else "on")
    ne, ee, _ = run(True, 0)
    d.set_yz_fold("auto"); d.set_tail_fusion("auto")
    ok = na == nb == nc == nd == ne
    for k in names:
        same = torch.equal(a[k], b[k]) and torch.equal(a[k], cc[k]) and torch.equal(a[k], dd[k]) and torch.equal(a[k], ee[k])
        fin = bool(torch.isfinite(a[k]).all())
        if not (same and fin):
            print(name, k, "DIFFERENT" if not same else "", "NON-FINITE" if not fin else "")
        ok = ok and same and fin
    print("%s: %d sub-steps, rows flagged in the last stage %d of %d, max|w| %.3f m/s: %s" % (
        name, na, rows[0], rows[1], float(a["wvel"].abs().max()), "bit-identical x5" if ok else "FAILED"), flush=True)
    allok = allok and ok
    d.finalize(c); c.dm.finalize(); del c, d, a, b, cc, dd, ee, init
    torch.cuda.empty_cache()
print("SOAK OK" if allok else "SOAK FAILED")
sys.exit(0 if allok else 1)
