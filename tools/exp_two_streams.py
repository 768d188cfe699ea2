"""Experiment: does running two half-ensembles on two HIP streams overlap the HBM-bound update kernel of one half
with the VALU-bound flux kernel of the other?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pam_amd import Dycore, PamCoupler, idealized as idz

def make(nens, stream, nx=32, ny=32, nz=60):
    zint = idz.l60_interfaces()
    with torch.cuda.stream(stream):
        c = PamCoupler("cuda:0"); c.set_option("crm_dt", 2.0)
        c.allocate_coupler_state(nz, ny, nx, nens); c.set_grid(nx*1000., ny*1000., zint)
        c.add_tracer("water_vapor", "", True, True)
        d = Dycore(); d.init(c)
        f = idz.supercell_fields(16, nx, ny, nz, zint, magnitude=0.1)
        reps = nens // 16
        for k in ("density_dry","uvel","vvel","wvel","temp"):
            c.dm.get(k).copy_(torch.from_numpy(f[k]).cuda().repeat(1,1,1,reps))
        c.dm.get("water_vapor").copy_(torch.from_numpy(f["tracers"][0]).cuda().repeat(1,1,1,reps))
        d.declare_current_profile_as_hydrostatic(c)
    torch.cuda.synchronize()
    return c, d

def run(parts, steps=3):
    dts = [d.compute_time_step(c) for c, d, s in parts]
    dt = min(dts)
    for c, d, s in parts:
        with torch.cuda.stream(s): d.timeStep(c, dt_dyn_hint=dt)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); sub = 0
    for _ in range(steps):
        for c, d, s in parts:
            with torch.cuda.stream(s): sub += d.timeStep(c, dt_dyn_hint=dt) * c.get_nens()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return sub * 60*32*32 / el / 1e9

s0 = torch.cuda.current_stream()
c, d = make(1024, s0); print("1 x 1024 on one stream : %.3f G cell-updates/s" % run([(c, d, s0)])); d.finalize(c); del c, d
torch.cuda.empty_cache()
for nparts in (2, 4):
    streams = [torch.cuda.Stream() for _ in range(nparts)]
    parts = []
    for s in streams:
        c, d = make(1024 // nparts, s); parts.append((c, d, s))
    print("%d x %d on %d streams   : %.3f G cell-updates/s" % (nparts, 1024//nparts, nparts, run(parts)))
    for c, d, s in parts: d.finalize(c)
    del parts; torch.cuda.empty_cache()
