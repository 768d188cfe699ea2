"""Timing of the coupler modules around the dycore at BASELINE sizes (HIP events around the C-ABI calls)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pam_amd import PamCoupler, Microphysics, modules, idealized as idz

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for label, nens, nx, ny in (("C3 grid 4096 x 32x1x60", 4096, 32, 1), ("C2 grid 1024 x 32x32x60", 1024, 32, 32)):
    nz = 60
    zint = idz.l60_interfaces()
    c = PamCoupler("cuda:0"); c.set_option("crm_dt", 2.0); c.set_option("gcm_physics_dt", 1200.0)
    c.allocate_coupler_state(nz, ny, nx, nens); c.set_grid(nx * 1000.0, max(ny, 1) * 1000.0, zint)
    micro = Microphysics(); micro.init(c)
    f = idz.supercell_fields(16, nx, ny, nz, zint, tracers=(("water_vapor", True, True),), magnitude=0.5)
    dm = c.get_data_manager_device_readwrite()
    rep = nens // 16
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp"):
        dm.get(k).copy_(torch.from_numpy(f[k]).to("cuda:0").repeat(1, 1, 1, rep))
    dm.get("water_vapor").copy_(torch.from_numpy(f["tracers"][0]).to("cuda:0").repeat(1, 1, 1, rep) * 1.3)
    dm.get("precip_liquid").copy_(dm.get("density_dry") * 1e-3)
    cells = nens * nx * ny * nz
    t = timeit(lambda: micro.timeStep(c))
    # 11 doubles read + 4 written per cell and sub-cycle in the column kernel, 6 + 5 in the prep kernel
    print("%s: Kessler timeStep %.3f ms  (%.1f G cells/s, ~%.2f TB/s of algorithmic traffic at 1 sub-cycle)" %
          (label, t, cells / t / 1e6, cells * (26 * 8) / t / 1e9))
    t = timeit(lambda: modules.sponge_layer(c))
    print("%s: sponge_layer %.3f ms" % (label, t))
    del c, micro, dm
    torch.cuda.empty_cache()


# GCM forcing (compute once per GCM step, apply once per CRM step) at the C2 grid
def _gcm():
    nens, nx, ny, nz = 1024, 32, 32, 60
    zint = idz.l60_interfaces()
    c = PamCoupler("cuda:0"); c.set_option("crm_dt", 2.0); c.set_option("gcm_physics_dt", 1200.0)
    c.allocate_coupler_state(nz, ny, nx, nens); c.set_grid(nx * 1000.0, ny * 1000.0, zint)
    for n in ("water_vapor", "cloud_water", "ice", "cloud_water_num", "ice_num", "rain_num"):
        c.add_tracer(n, "", True, n in ("water_vapor", "cloud_water", "ice"))
    f = idz.supercell_fields(16, nx, ny, nz, zint, magnitude=0.5)
    dm = c.get_data_manager_device_readwrite()
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp"):
        dm.get(k).copy_(torch.from_numpy(f[k]).to("cuda:0").repeat(1, 1, 1, nens // 16))
    dm.get("water_vapor").copy_(torch.from_numpy(f["tracers"][0]).to("cuda:0").repeat(1, 1, 1, nens // 16))
    for k in ("gcm_density_dry", "gcm_temp", "gcm_water_vapor"):
        src = {"gcm_density_dry": "density_dry", "gcm_temp": "temp", "gcm_water_vapor": "water_vapor"}[k]
        dm.get(k).copy_(dm.get(src).mean(dim=(1, 2)) * 1.01)
    modules.compute_gcm_forcing_tendencies(c)
    print("C2 grid: compute_gcm_forcing_tendencies %.3f ms, apply_gcm_forcing_tendencies %.3f ms" %
          (timeit(lambda: modules.compute_gcm_forcing_tendencies(c), 3), timeit(lambda: modules.apply_gcm_forcing_tendencies(c), 3)))


_gcm()
