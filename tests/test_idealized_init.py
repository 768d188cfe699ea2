"""Dycore::init's optional idealised initial data (row a15 of SURVEY.md section 8a; Dycore.h:986-1090): the 9-point-GLL
`thermal` bubble (:1021-1088) and `init_supercell` (:1096-1276), oracle properties on CPU and HIP-vs-oracle on the GPU."""
import numpy as np
import pytest

from oracle import awfl_oracle as ao
from pam_amd import idealized as idz


def _oracle(nens, nx, ny, nz, zint, xlen, ylen, tr=idz.TRACERS_NONE):
    names, pos, mass, idwv = idz.tracer_flags(tr)
    o = ao.OracleDycore(nens, nx, ny, nz, xlen, ylen, np.diff(zint), pos, mass, idwv)
    f = {k: np.full((nz, ny, nx, nens), np.nan) for k in ("density_dry", "uvel", "vvel", "wvel", "temp")}
    f["tracers"] = np.full((len(tr), nz, ny, nx, nens), np.nan)
    return o, f


def test_oracle_thermal_init_properties():
    nens, nx, ny, nz = 2, 12, 1, 16
    zint = idz.uniform_interfaces(nz, 8000.0)
    o, f = _oracle(nens, nx, ny, nz, zint, 12000.0, 12000.0)
    o.init_idealized(f, "thermal", zint)
    assert all(np.isfinite(v).all() for v in f.values())
    assert np.all(f["uvel"] == 0) and np.all(f["wvel"] == 0) and np.all(f["tracers"] == 0)
    # warm bubble centred at (xlen/2, 2000 m), radius 2000 m: warmest cell is the one containing the centre; far cells
    # carry the unperturbed theta=300 K profile (T = 300 * exner)
    k, j, i, e = np.unravel_index(np.argmax(f["temp"] - f["temp"][:, :, :1, :]), f["temp"].shape)
    assert abs((i + 0.5) * 1000.0 - 6000.0) <= 1000.0 and abs(0.5 * (zint[k] + zint[k + 1]) - 2000.0) <= 500.0
    # theta' = 2 K at constant density: T'/T = gamma theta'/theta -> the cell-averaged peak is ~2 K in T as well
    assert 1.0 < (f["temp"] - f["temp"][:, :, :1, :]).max() < 2.8
    # the column is hydrostatic: the dycore's variable gravity for it is 9.81 to the quadrature/WENO truncation error
    o.declare_current_profile_as_hydrostatic({**f, "temp": np.ascontiguousarray(np.broadcast_to(f["temp"][:, :, :1, :], f["temp"].shape))})
    assert np.allclose(o.variable_gravity, 9.81, atol=2e-3)


def test_oracle_supercell_init_properties():
    nens, nx, ny, nz = 2, 6, 4, 30
    zint = idz.uniform_interfaces(nz, 18000.0)
    tr = idz.TRACERS_KESSLER_SHOC
    o, f = _oracle(nens, nx, ny, nz, zint, 6000.0, 4000.0, tr)
    o.init_idealized(f, "supercell", zint)
    assert all(np.isfinite(v).all() for v in f.values())
    # horizontally uniform; sheared zonal wind from -15 to +15 m/s; vapour mixing ratio capped at 14 g/kg
    for k in ("density_dry", "uvel", "temp"):
        assert np.allclose(f[k], f[k][:, :1, :1, :], rtol=1e-14, atol=1e-14)
    assert abs(f["uvel"][0].min() + 15 - 30 * 300.0 / 5000.0) < 0.1 and np.allclose(f["uvel"][-1], 15.0)
    qv = f["tracers"][0] / f["density_dry"]
    assert 0.0135 < qv.max() <= 0.014 + 1e-12 and np.all(f["tracers"][1:] == 0)
    # the independent numpy sounding of pam_amd.idealized (supercell_init.h, 5-point GLL) agrees to quadrature accuracy
    rho_d, u, v, w, T, rho_v = idz.supercell_column(zint)
    assert np.allclose(f["temp"][:, 0, 0, 0], T, rtol=2e-3) and np.allclose(f["density_dry"][:, 0, 0, 0], rho_d, rtol=5e-3)
    with pytest.raises(ValueError):
        o.init_idealized(f, "bogus", zint)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,ny", [("thermal", 1), ("thermal", 5), ("supercell", 1), ("supercell", 4)])
def test_gpu_idealized_init_matches_oracle(kind, ny, tmp_path):
    import torch
    from pam_amd import Dycore, PamCoupler, PamAmdError
    nens, nx, nz = 70, 7, 14
    tr = idz.TRACERS_KESSLER_SHOC
    zint = idz.stretched_interfaces(nz, 14000.0, ratio=1.1)
    xlen, ylen = nx * 1000.0, (ny if ny > 1 else nx) * 1000.0
    yml = tmp_path / "input.yaml"
    yml.write_text("initData: %s\n" % kind)
    coupler = PamCoupler("cuda:0")
    coupler.set_option("crm_dt", 2.0)
    coupler.set_option("standalone_input_file", str(yml))      # driver.cpp:126-128
    coupler.allocate_coupler_state(nz, ny, nx, nens)
    coupler.set_grid(xlen, ylen, zint)
    for n, p, m in tr:
        coupler.add_tracer(n, "", p, m)
    dycore = Dycore()
    dycore.init(coupler)                                         # fills the coupler state from the YAML's initData
    torch.cuda.synchronize()
    got = coupler.dump_fields()
    o, f = _oracle(nens, nx, ny, nz, zint, xlen, ylen, tr)
    o.init_idealized(f, kind, zint)
    for k in got:
        assert np.abs(got[k] - f[k]).max() <= 1e-13 * max(np.abs(f[k]).max(), 1e-300), k
    # and the state is usable: one step runs and stays finite
    dycore.declare_current_profile_as_hydrostatic(coupler)
    dycore.timeStep(coupler)
    torch.cuda.synchronize()
    assert all(np.isfinite(v).all() for v in coupler.dump_fields().values())
    with pytest.raises(PamAmdError):
        dycore.init_idealized(coupler, "bogus")
    dycore.finalize(coupler)
