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
@pytest.mark.parametrize("ny", [1, 4])
def test_gpu_sponge_layer_matches_oracle(ny):
    import torch
    from pam_amd import PamCoupler, modules
    tr = idz.TRACERS_KESSLER_SHOC
    nens, nx, nz = 70, 5, 12
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
