"""Coupler modules around the dycore ("next row" N2): modules::sponge_layer
(pam_core/modules/sponge_layer.h:8-95), oracle properties on CPU and HIP-vs-oracle parity on the GPU."""
import copy

import numpy as np
import pytest

from oracle import awfl_oracle as ao
from pam_amd import idealized as idz


def _case(nens=3, nx=5, ny=4, nz=12, tr=idz.TRACERS_KESSLER_SHOC):
    zint = idz.stretched_interfaces(nz, 12000.0)
    zi = zint[:, None] * (1 + 0.01 * np.arange(nens))[None, :]
    zm = 0.5 * (zi[:-1] + zi[1:])
    f = idz.supercell_fields(nens, nx, ny, nz, zint, tracers=tr, magnitude=1.0)
    idz.add_tracer_blobs(f, tr, nx * 500.0, ny * 500.0, zint)
    f["wvel"] += 0.3 * np.cos(np.arange(nx))[None, None, :, None]
    return zint, zi, zm, f


def test_oracle_sponge_layer_properties():
    zint, zi, zm, f = _case()
    g = copy.deepcopy(f)
    ao.sponge_layer(g, zi, zm, 2.0, num_layers=5, time_scale=60.0)
    # only the top 5 levels change; horizontal means of non-w fields are preserved; w is damped towards zero
    for k in ("density_dry", "uvel", "temp", "wvel", "tracers"):
        assert np.array_equal(g[k][..., :-5, :, :, :], f[k][..., :-5, :, :, :])
    for k in ("density_dry", "uvel", "vvel", "temp"):
        assert np.allclose(g[k][-5:].mean(axis=(1, 2)), f[k][-5:].mean(axis=(1, 2)), rtol=1e-14, atol=1e-14)
    assert np.all(np.abs(g["wvel"][-1]) < np.abs(f["wvel"][-1]) + 1e-300)
    # the relaxation factor is largest at the model top (cos profile, sponge_layer.h:89-91)
    d = np.abs(g["wvel"][-5:] - f["wvel"][-5:]) / np.maximum(np.abs(f["wvel"][-5:]), 1e-300)
    assert np.all(np.diff(d.max(axis=(1, 2, 3))) > 0)


@pytest.mark.gpu
@pytest.mark.parametrize("ny,nens,nx", [(1, 70, 5), (4, 70, 5), (6, 3, 11), (12, 1, 24)],
                         ids=["2d", "3d", "strips_small_ensemble", "strips_one_member"])
def test_gpu_sponge_layer_matches_oracle(ny, nens, nx):
    """the last two cases have ny*nx >= 16: the horizontal means are summed in STRIPS (sponge_mean_kernel + sponge_mean_finish_kernel)"""
    import torch
    from pam_amd import PamCoupler, modules
    tr = idz.TRACERS_KESSLER_SHOC
    nz = 12
    zint, zi, zm, f = _case(nens=nens, nx=nx, ny=ny, nz=nz, tr=tr)
    coupler = PamCoupler("cuda:0")
    coupler.set_option("crm_dt", 2.0)
    coupler.set_option("sponge_num_layers", 4)
    coupler.set_option("sponge_time_scale", 30.0)
    coupler.allocate_coupler_state(nz, ny, nx, nens)
    coupler.set_grid(nx * 500.0, nx * 500.0, zi)
    for n, p, m in tr:
        coupler.add_tracer(n, "", p, m)
    coupler.load_fields(f)
    dirty = coupler.run_module("sponge_layer", modules.sponge_layer)
    torch.cuda.synchronize()
    assert "temp" in dirty and "water_vapor" in dirty
    got = coupler.dump_fields()
    ao.sponge_layer(f, zi, zm, 2.0, num_layers=4, time_scale=30.0)
    for k in got:
        assert np.abs(got[k] - f[k]).max() <= 1e-14 * max(np.abs(f[k]).max(), 1e-300), k


@pytest.mark.gpu
def test_gpu_dry_crm_step_dycore_then_sponge():
    """The dry part of the driver loop (driver.cpp:248-250): dycore.timeStep then sponge_layer, twice."""
    import torch
    from pam_amd import Dycore, PamCoupler, modules
    tr = idz.TRACERS_NONE
    nens, nx, ny, nz = 4, 8, 1, 12
    names, pos, mass, idwv = idz.tracer_flags(tr)
    zint = idz.stretched_interfaces(nz, 12000.0)
    zi = zint[:, None] * np.ones((1, nens))
    zm = 0.5 * (zi[:-1] + zi[1:])
    f = idz.supercell_fields(nens, nx, ny, nz, zint, tracers=tr, magnitude=0.5)
    coupler = PamCoupler("cuda:0")
    coupler.set_option("crm_dt", 2.0)
    coupler.allocate_coupler_state(nz, ny, nx, nens)
    coupler.set_grid(nx * 500.0, nx * 500.0, zint)
    coupler.add_tracer("water_vapor", "", True, True)
    dycore = Dycore()
    dycore.init(coupler)
    coupler.load_fields(f)
    dycore.declare_current_profile_as_hydrostatic(coupler)
    o = ao.OracleDycore(nens, nx, ny, nz, nx * 500.0, nx * 500.0, np.diff(zint), pos, mass, idwv)
    o.declare_current_profile_as_hydrostatic(f)
    for _ in range(2):
        coupler.run_module("dycore", dycore.timeStep)
        coupler.run_module("sponge_layer", modules.sponge_layer)
        o.time_step(f, 2.0)
        ao.sponge_layer(f, zi, zm, 2.0)
    torch.cuda.synchronize()
    got = coupler.dump_fields()
    for k in ("density_dry", "temp"):
        assert np.abs(got[k] - f[k]).max() <= 1e-12 * np.abs(f[k]).max(), k
    for k in ("uvel", "wvel"):
        assert np.abs(got[k] - f[k]).max() <= 1e-9 * np.abs(f[k]).max(), k
    dycore.finalize(coupler)


# ---------------------------------------------------------------------------------------------------------------------
# "next row" N1: modules::compute_gcm_forcing_tendencies / apply_gcm_forcing_tendencies (pam_core/modules/gcm_forcing.h)
P3_TRACERS = idz.TRACERS_P3_SHOC


def _gcm_case(nens=3, nx=5, ny=2, nz=8, seed=1, starve_liquid=False, starve_level=False):
    rng = np.random.default_rng(seed)
    crm = {n: np.ascontiguousarray(rng.uniform(0.5, 1.5, (nz, ny, nx, nens))) for n in ao.GCM_FORCING_CRM}
    for n in ("water_vapor", "cloud_water", "ice"):
        crm[n] *= 0.01
    crm["ice"][nz // 2:] = 0.0
    crm["temp"] += 270.0
    if starve_level:       # every other column has almost no liquid
        crm["cloud_water"][:, :, ::2] *= 0.02
    # GCM state close to the CRM's column means (a weak forcing that leaves every cell positive) ...
    gcm = {g: np.ascontiguousarray(crm[c].mean(axis=(1, 2)) * rng.uniform(0.9, 1.1, (nz, nens)))
           for g, c in zip(ao.GCM_FORCING_GCM, ao.GCM_FORCING_CRM)}
    if starve_level:       # ... or asking for half the liquid: the poor columns go negative, their level can pay
        gcm["gcm_cloud_water"] *= 0.5
    if starve_liquid:      # ... or for less than nothing at one level: that level cannot pay -> whole-CRM fallback.  (A GCM
        # column with no liquid anywhere would end in 0/0 = NaN in the reference's fallback, gcm_forcing.h:270.)
        gcm["gcm_cloud_water"][1] *= -0.2
    dz = np.ascontiguousarray(rng.uniform(50.0, 400.0, (nz, nens)))
    return crm, gcm, dz


def test_oracle_gcm_forcing_relaxes_column_means_to_the_gcm_state():
    crm, gcm, dz = _gcm_case()
    dt_gcm, crm_dt = 1200.0, 300.0
    tend = ao.compute_gcm_forcing_tendencies(crm, gcm, dt_gcm)
    assert np.allclose(tend["gcm_forcing_tend_qtot"], tend["gcm_forcing_tend_qv"] + tend["gcm_forcing_tend_ql"] +
                       tend["gcm_forcing_tend_qi"], rtol=0, atol=1e-18)
    for _ in range(4):     # forcing alone over one GCM step: state_crm_new == state_gcm (gcm_forcing.h:9-16)
        ao.apply_gcm_forcing_tendencies(crm, gcm, tend, dz, crm_dt, dt_gcm)
    for a, b in (("density_dry", "gcm_density_dry"), ("uvel", "gcm_uvel"), ("vvel", "gcm_vvel"), ("temp", "gcm_temp"),
                 ("cloud_water_num", "gcm_num_liq")):
        assert np.abs(crm[a].mean(axis=(1, 2)) - gcm[b]).max() < 1e-12 * np.abs(gcm[b]).max(), a
    # water is forced as a mixing ratio w.r.t. (rho_d + rho_v), which is not linear in the densities: the diagnosed
    # density forcing left after the last application is small but not zero
    assert np.abs(tend["gcm_forcing_tend_rho_v"]).max() * dt_gcm < 0.2 * gcm["gcm_water_vapor"].max()


GCM_CASES = [({}, 0), ({"starve_level": True}, 2), ({"starve_liquid": True}, 2 | 32)]   # union over the 4 applications


@pytest.mark.parametrize("kw,want", GCM_CASES)
def test_oracle_gcm_forcing_hole_filling_paths(kw, want):
    crm, gcm, dz = _gcm_case(**kw)
    tend = ao.compute_gcm_forcing_tendencies(crm, gcm, 1200.0)
    mask = 0
    for _ in range(4):
        mask |= ao.apply_gcm_forcing_tendencies(crm, gcm, tend, dz, 300.0, 1200.0)
    assert mask == want
    for n in ("water_vapor", "cloud_water", "ice", "cloud_water_num", "ice_num", "rain_num"):
        assert crm[n].min() >= 0.0 and np.all(np.isfinite(crm[n])), n


def _gcm_gpu_coupler(crm, gcm, dz, dt_gcm, crm_dt):
    import torch
    from pam_amd import PamCoupler
    nz, ny, nx, nens = crm["density_dry"].shape
    coupler = PamCoupler("cuda:0")
    coupler.set_option("crm_dt", crm_dt)
    coupler.set_option("gcm_physics_dt", dt_gcm)
    coupler.allocate_coupler_state(nz, ny, nx, nens)
    zint = np.concatenate([np.zeros((1, nens)), np.cumsum(dz, axis=0)], axis=0)
    coupler.set_grid(nx * 500.0, ny * 500.0, zint)
    for n, p, m in P3_TRACERS:      # registration order of physics/micro/p3/Microphysics.h:119-127 + SHOC's tke
        coupler.add_tracer(n, "", p, m)
    dm = coupler.get_data_manager_device_readwrite()
    for n, a in list(crm.items()) + list(gcm.items()):
        dm.get(n).copy_(torch.from_numpy(a))
    return coupler, dm, zint


@pytest.mark.gpu
@pytest.mark.parametrize("kw,want", GCM_CASES)
def test_gpu_gcm_forcing_matches_oracle(kw, want):
    import torch
    from pam_amd import modules
    crm, gcm, dz = _gcm_case(nens=70, nx=6, ny=3, nz=9, **kw)
    dt_gcm, crm_dt = 1200.0, 300.0
    coupler, dm, zint = _gcm_gpu_coupler(crm, gcm, dz, dt_gcm, crm_dt)
    dz_used = np.diff(zint, axis=0)       # what set_grid stored (cumsum round trip)
    dirty = coupler.run_module("compute_gcm_forcing_tendencies", modules.compute_gcm_forcing_tendencies)
    assert "gcm_forcing_tend_uvel" in dirty and "density_dry" not in dirty
    tend = ao.compute_gcm_forcing_tendencies(crm, gcm, dt_gcm)
    for n in tend:
        if n[-5:] in ("rho_v", "rho_l", "rho_i"):
            continue
        got = dm.get(n, readonly=True).cpu().numpy()
        # a tendency is (gcm - mean)/dt_gcm: its round-off scales with the state, not with the (small) difference
        scale = max(np.abs(g).max() for g in gcm.values()) / dt_gcm
        assert np.abs(got - tend[n]).max() <= 1e-14 * scale, n
    mask = 0
    for _ in range(4):
        out = {}
        coupler.run_module("apply_gcm_forcing_tendencies", lambda c: out.setdefault("m", modules.apply_gcm_forcing_tendencies(c)))
        mask |= out["m"]
        mref = ao.apply_gcm_forcing_tendencies(crm, gcm, tend, dz_used, crm_dt, dt_gcm)
        assert out["m"] == mref
    torch.cuda.synchronize()
    assert mask == want
    for n in ao.GCM_FORCING_CRM:
        got = dm.get(n, readonly=True).cpu().numpy()
        assert np.abs(got - crm[n]).max() <= 1e-12 * max(np.abs(crm[n]).max(), 1e-300), n
    for n in ("gcm_forcing_tend_rho_v", "gcm_forcing_tend_rho_l", "gcm_forcing_tend_rho_i"):
        got = dm.get(n, readonly=True).cpu().numpy()
        assert np.abs(got - tend[n]).max() <= 1e-13 * 0.015 / dt_gcm, n


# ---------------------------------------------------------------------------------------------------------------------
# rest of "next row" N2: broadcast_initial_gcm_column, perturb_temperature
def test_oracle_perturb_temperature_shape():
    nz, ny, nx, nens = 16, 3, 5, 4
    T0 = np.full((nz, ny, nx, nens), 300.0) + np.arange(nz)[:, None, None, None]
    T = T0.copy()
    ao.perturb_temperature(T, np.arange(nens) + 7, 0.1)
    nl = nz // 4
    assert np.array_equal(T[nl:], T0[nl:])                                         # only the lowest nz/4 levels
    d = np.abs(T[:nl] - T0[:nl]).max(axis=(1, 2, 3))
    assert np.all(np.diff(d) < 0) and 0.05 < d[0] <= 0.1 * 1.2                     # decays with height, bounded by magnitude
    assert np.abs(T[:nl].mean(axis=(1, 2)) - T0[:nl].mean(axis=(1, 2))).max() < 1e-12   # horizontal mean restored
    assert not np.array_equal(T[0, :, :, 0], T[0, :, :, 1])                        # members draw different numbers
    T2 = T0.copy()
    idz.perturb_temperature(T2, 0.1, id0=7)                                        # the numpy generator of the bench inputs
    assert np.abs(T - T2).max() < 1e-12


@pytest.mark.gpu
def test_gpu_broadcast_and_perturb_match_oracle():
    import torch
    from pam_amd import PamCoupler, modules
    nens, nx, ny, nz = 70, 5, 3, 17
    rng = np.random.default_rng(3)
    coupler = PamCoupler("cuda:0")
    coupler.allocate_coupler_state(nz, ny, nx, nens)
    coupler.set_grid(nx * 500.0, ny * 500.0, np.linspace(0, 12000.0, nz + 1))
    coupler.add_tracer("water_vapor", "", True, True)
    dm = coupler.get_data_manager_device_readwrite()
    gcm = {n: np.ascontiguousarray(rng.uniform(0.5, 1.5, (nz, nens))) for n in ao.BROADCAST_GCM}
    gcm["gcm_temp"] += 280.0
    for n, a in gcm.items():
        dm.get(n).copy_(torch.from_numpy(a))
    crm = {n: np.zeros((nz, ny, nx, nens)) for n in ao.BROADCAST_CRM}
    dirty = coupler.run_module("broadcast", modules.broadcast_initial_gcm_column_dry_density)
    assert "density_dry" in dirty and "temp" not in dirty
    ao.broadcast_initial_gcm_column(crm, gcm, dry_density_only=True)
    for n in ao.BROADCAST_CRM:
        assert np.array_equal(dm.get(n, readonly=True).cpu().numpy(), crm[n]), n
    coupler.run_module("broadcast", modules.broadcast_initial_gcm_column)
    ao.broadcast_initial_gcm_column(crm, gcm)
    for n in ao.BROADCAST_CRM:
        assert np.array_equal(dm.get(n, readonly=True).cpu().numpy(), crm[n]), n
    ids = np.arange(nens) * 3 + 11
    keep = modules.perturb_temperature(coupler, ids, 0.25)
    torch.cuda.synchronize()
    ao.perturb_temperature(crm["temp"], ids, 0.25)
    got = dm.get("temp", readonly=True).cpu().numpy()
    assert np.abs(got - crm["temp"]).max() <= 1e-14 * crm["temp"].max()
    assert np.abs(got - gcm["gcm_temp"][:, None, None, :]).max() > 0.1
    with pytest.raises(Exception):
        modules.perturb_temperature(coupler, ids[:-1], 0.25)


@pytest.mark.gpu
@pytest.mark.parametrize("grid", ["L60", "uniform40"])
def test_driver_supercell_column_matches_oracle(grid):
    """supercell_init of the standalone driver (standalone/mmf_simplified/supercell_init.h:7-135) as a device kernel against
    the oracle's restatement: exp/pow of the device libm vs glibc -> 1e-13 relative to the column maximum."""
    import torch
    from oracle import awfl_oracle as ao
    from pam_amd import modules
    zint = idz.l60_interfaces() if grid == "L60" else idz.uniform_interfaces(40, 20000.0)
    c = idz.CONSTS_DEFAULT
    got = modules.supercell_init(torch.from_numpy(np.ascontiguousarray(zint)).to("cuda:0"), c["R_d"], c["R_v"], c["grav"])
    torch.cuda.synchronize()
    exp = ao.supercell_init(zint, c)
    for name, g, e in zip(("rho_d", "uvel", "vvel", "wvel", "temp", "rho_v"), got, exp):
        g = g.cpu().numpy()
        assert np.isfinite(g).all(), name
        assert np.abs(g - e).max() <= 1e-13 * max(np.abs(e).max(), 1e-300), (name, np.abs(g - e).max())
    assert exp[5].max() > 0 and exp[0][0] > 1.0      # a moist, ground-based column


@pytest.mark.gpu
def test_gpu_sponge_layer_does_not_depend_on_how_the_ensemble_is_sharded():
    """the strips of the horizontal means follow from (nx, ny) alone, so a member's result is bit-identical whether the call
    holds the whole ensemble or a shard of it (1-GPU run vs members sharded over N GPUs)"""
    import torch
    from pam_amd import PamCoupler, modules
    tr = idz.TRACERS_NONE
    nens, nx, ny, nz = 96, 12, 10, 12
    zint, zi, zm, f = _case(nens=nens, nx=nx, ny=ny, nz=nz, tr=tr)

    def run(lo, hi):
        coupler = PamCoupler("cuda:0")
        coupler.set_option("crm_dt", 2.0)
        coupler.allocate_coupler_state(nz, ny, nx, hi - lo)
        coupler.set_grid(nx * 500.0, ny * 500.0, zi[:, lo:hi])
        coupler.add_tracer("water_vapor", "", True, True)
        coupler.load_fields({k: np.ascontiguousarray(v[..., lo:hi]) for k, v in f.items()})
        coupler.run_module("sponge_layer", modules.sponge_layer)
        torch.cuda.synchronize()
        return coupler.dump_fields()
    whole = run(0, nens)
    for lo, hi in ((0, 5), (5, 96), (40, 41)):
        part = run(lo, hi)
        for k in whole:
            assert np.array_equal(whole[k][..., lo:hi], part[k]), (k, lo, hi)
