import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pam_amd import Dycore, PamCoupler, idealized as idz
nens, nx, ny, nz = 4, 9, 4, 9
tr = idz.TRACERS_KESSLER_SHOC
zint = idz.stretched_interfaces(nz, 12000.0)
xlen, ylen = nx*500., ny*500.
f = idz.supercell_fields(nens, nx, ny, nz, zint, tracers=tr, magnitude=0.5); idz.add_tracer_blobs(f, tr, xlen, ylen, zint)
out = {}
for seg in (8, 1, 2, 3, 16):
    c = PamCoupler("cuda:0"); c.set_option("crm_dt", 2.0); c.allocate_coupler_state(nz, ny, nx, nens); c.set_grid(xlen, ylen, zint)
    for n, p, m in tr: c.add_tracer(n, "", p, m)
    d = Dycore(); d.init(c); d.set_flux_segment(seg); c.load_fields(f)
    d.declare_current_profile_as_hydrostatic(c); d.convert_coupler_to_dynamics(c); d.debug_flux_stage(1.0); torch.cuda.synchronize()
    nt = len(tr)
    out[seg] = [d.debug_buffer("flux_x").cpu().numpy().reshape(5+nt, nz, ny, nx, nens).copy(),
                d.debug_buffer("flux_y").cpu().numpy().reshape(5+nt, nz, ny, nx, nens).copy(),
                d.debug_buffer("flux_z").cpu().numpy().reshape(5+nt, nz+1, ny, nx, nens).copy()]
    d.finalize(c)
    if seg != 8:
        for di, nm in enumerate("xyz"):
            a, b = out[8][di], out[seg][di]
            diff = np.abs(a-b)
            bad = np.argwhere(diff > 1e-13*np.abs(a).max())
            print("seg", seg, "dir", nm, "maxdiff", diff.max(), "nbad", len(bad), "first bad (l,k,j,i,e):", bad[:4].tolist())
