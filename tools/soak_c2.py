"""Race check at full C2 size: N timeSteps with the shipped overlapped schedule (ensemble ranges on prioritised streams,
event-chained flux kernels) must give bit-identical coupler fields to the same N steps run as one range on one stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pam_amd import Dycore, PamCoupler, idealized as idz

nens, nx, ny, nz = 1024, 32, 32, 60
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
zint = idz.l60_interfaces()
tr = idz.TRACERS_NONE
f = idz.supercell_fields(16, nx, ny, nz, zint, tracers=tr, magnitude=0.5)
c = PamCoupler("cuda:0"); c.set_option("crm_dt", 2.0)
c.allocate_coupler_state(nz, ny, nx, nens); c.set_grid(nx * 1000.0, ny * 1000.0, zint)
for n, p, m in tr: c.add_tracer(n, "", p, m)
d = Dycore(); d.init(c)
names = ["density_dry", "uvel", "vvel", "wvel", "temp"] + c.get_tracer_names()
init = {}
for k in names[:5]:
    init[k] = torch.from_numpy(f[k]).to("cuda:0").repeat(1, 1, 1, nens // 16).contiguous()
init["temp"] += (torch.arange(nens, device="cuda:0", dtype=torch.float64) // 16) * 1e-3
for t, n in enumerate(c.get_tracer_names()):
    init[n] = torch.from_numpy(f["tracers"][t]).to("cuda:0").repeat(1, 1, 1, nens // 16).contiguous()

def run(chunks):
    for k in names: c.dm.get(k).copy_(init[k])
    d.set_ensemble_chunks(chunks)
    d.declare_current_profile_as_hydrostatic(c)
    tot = 0
    for _ in range(nsteps): tot += d.timeStep(c)
    torch.cuda.synchronize()
    return tot, {k: c.dm.get(k, readonly=True).clone() for k in names}

na, a = run(0)
nb, b = run(1)
ok = na == nb
for k in names:
    same = torch.equal(a[k], b[k])
    fin = bool(torch.isfinite(a[k]).all())
    print(k, "bit-identical" if same else "DIFFERENT max|d|=%g" % float((a[k] - b[k]).abs().max()), "finite" if fin else "NON-FINITE")
    ok = ok and same and fin
print("substeps", na, nb, "max|w| = %.3f m/s" % float(a["wvel"].abs().max()))
print("SOAK OK" if ok else "SOAK FAILED")
sys.exit(0 if ok else 1)
