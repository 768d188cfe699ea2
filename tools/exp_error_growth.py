"""Growth of the HIP-vs-oracle difference with the number of sub-steps (documentation of the parity budget)."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pam_amd import Dycore, PamCoupler, idealized as idz
from oracle import awfl_oracle as ao
nens, nx, ny, nz = 4, 16, 8, 20
tr = idz.TRACERS_KESSLER_SHOC
names, pos, mass, idwv = idz.tracer_flags(tr)
zint = idz.stretched_interfaces(nz, 15000.0, ratio=1.08)
xlen, ylen = nx*1000., ny*1000.
f = idz.supercell_fields(nens, nx, ny, nz, zint, tracers=tr, magnitude=0.5); idz.add_tracer_blobs(f, tr, xlen, ylen, zint)
fo = copy.deepcopy(f)
c = PamCoupler("cuda:0"); c.set_option("crm_dt", 4.0); c.allocate_coupler_state(nz, ny, nx, nens); c.set_grid(xlen, ylen, zint)
for n, p, m in tr: c.add_tracer(n, "", p, m)
d = Dycore(); d.init(c); c.load_fields(f); d.declare_current_profile_as_hydrostatic(c)
o = ao.OracleDycore(nens, nx, ny, nz, xlen, ylen, np.diff(zint), pos, mass, idwv); o.declare_current_profile_as_hydrostatic(fo)
sub = 0
for step in range(1, 41):
    n = d.timeStep(c); n2, _ = o.time_step(fo, 4.0); assert n == n2; sub += n
    if step in (1, 2, 5, 10, 20, 40):
        g = c.dump_fields()
        e = {k: float(np.abs(g[k]-fo[k]).max()/max(np.abs(fo[k]).max(), 1e-300)) for k in ("density_dry", "temp", "uvel", "vvel", "wvel")}
        print("timeSteps %3d sub-steps %4d max|w|=%.3f :" % (step, sub, np.abs(fo["wvel"]).max()), " ".join("%s=%.1e" % kv for kv in e.items()))
