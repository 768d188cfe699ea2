"""How launch-bound are the small configs?  host enqueue time of one timeStep vs its GPU time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pam_amd import Dycore, PamCoupler, idealized as idz
for (nens, nx, ny, tr, chunks) in ((512, 32, 1, idz.TRACERS_P3_SHOC, 0), (512, 32, 1, idz.TRACERS_P3_SHOC, 1),
                                  (4096, 32, 1, idz.TRACERS_KESSLER_SHOC, 0), (128, 32, 32, idz.TRACERS_NONE, 1)):
    nz = 60; zint = idz.l60_interfaces()
    c = PamCoupler("cuda:0"); c.set_option("crm_dt", 2.0); c.allocate_coupler_state(nz, ny, nx, nens); c.set_grid(nx*1000., nx*1000., zint)
    for n, p, m in tr: c.add_tracer(n, "", p, m)
    d = Dycore(); d.init(c); d.set_ensemble_chunks(chunks)
    f = idz.supercell_fields(16, nx, ny, nz, zint, tracers=tr, magnitude=0.1)
    if len(tr) > 1: idz.add_tracer_blobs(f, tr, nx*1000., nx*1000., zint)
    reps = nens // 16
    for k in ("density_dry","uvel","vvel","wvel","temp"): c.dm.get(k).copy_(torch.from_numpy(f[k]).cuda().repeat(1,1,1,reps))
    for t, n in enumerate(c.get_tracer_names()): c.dm.get(n).copy_(torch.from_numpy(f["tracers"][t]).cuda().repeat(1,1,1,reps))
    d.declare_current_profile_as_hydrostatic(c)
    dt = d.compute_time_step(c)
    for _ in range(2): d.timeStep(c, dt_dyn_hint=dt)
    torch.cuda.synchronize()
    host, tot = [], []
    for _ in range(5):
        t0 = time.perf_counter(); n = d.timeStep(c, dt_dyn_hint=dt); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append(t1 - t0); tot.append(t2 - t0)
    print("nens %d %dx%d NT=%d chunks=%d: substeps %d, host enqueue %.2f ms, total %.2f ms" % (nens, nx, ny, len(tr), chunks, n, 1e3*min(host), 1e3*min(tot)))
    d.finalize(c)
