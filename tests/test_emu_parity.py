"""CPU-only: the HIP kernel BODIES (pam_amd/csrc/awfl_device.h), compiled by g++ into a host emulation harness
(tests/emu/, test infrastructure), against the oracle.  This covers the host-visible logic -- launch geometry, segment
handling, periodic wrap, vertical ghosts, FCT multipliers and the periodic-seam quirk, SSPRK3 in-place aliasing, the
ensemble-uniform / per-member vertical tables -- on a machine without a GPU.  It checks INDEXING AND STRUCTURE, not the device's
rounding: the host build divides exactly where the device uses v_rcp_f64 + Newton steps (fast_rcp, weno_rcp in awfl_device.h) and
contracts nothing; only pow_pos_fast is the same arithmetic on both sides (tests/test_pow_pos.py).  The real device build is tested
by tests/test_gpu_parity.py and tests/test_kernel_level_parity.py (-m gpu)."""
import copy

import numpy as np
import pytest

from oracle import awfl_oracle as ao
from pam_amd import idealized as idz
import emu_harness as eh


def _rel(a, b):
    s = np.abs(b).max()
    return np.abs(a - b).max() / (s if s > 0 else 1.0)


CASES = {
    "2d_nt1_uniform_A": (3, 8, 1, 10, idz.TRACERS_NONE, idz.uniform_interfaces(10, 10000.0), {}, True, 8),
    "3d_nt1_stretched_A_seg3": (2, 7, 5, 9, idz.TRACERS_NONE, idz.stretched_interfaces(9, 12000.0), {}, True, 3),
    "3d_nt4_stretched_B": (2, 6, 6, 8, idz.TRACERS_KESSLER_SHOC, idz.stretched_interfaces(8, 12000.0), {}, False, 8),
    "3d_nt10_perens_A_p3": (3, 6, 4, 8, idz.TRACERS_P3_SHOC, idz.stretched_interfaces(8, 12000.0),
                            dict(per_ens=True, consts=idz.CONSTS_P3), True, 5),
    # one tracer (water_vapor) limited in every stage: the fix-up pass (tracer_fixup_line_body),
    # mode A and mode B; and mode B with one smooth tracer
    "3d_nt1_vapour_limited_A": (6, 8, 5, 9, idz.TRACERS_NONE, idz.stretched_interfaces(9, 12000.0), dict(dry_air=True), True, 8),
    "3d_nt1_vapour_limited_B": (5, 6, 4, 8, idz.TRACERS_NONE, idz.stretched_interfaces(8, 12000.0), dict(dry_air=True), False, 3),
    "3d_nt1_stretched_B": (2, 7, 5, 9, idz.TRACERS_NONE, idz.stretched_interfaces(9, 12000.0), {}, False, 8),
}


def _inputs(nens, nx, ny, nz, tr, zint, kw, consts, xlen, ylen):
    f = idz.supercell_fields(nens, nx, ny, nz, zint, consts=consts, tracers=tr, magnitude=1.0)
    idz.add_tracer_blobs(f, tr, xlen, ylen, zint)
    if kw.get("dry_air"):
        f["uvel"] -= 25.0
        f["vvel"] += 7.0
        idz.carve_dry_air(f, tr)
    return f


@pytest.mark.parametrize("fused", [False, True], ids=["three_kernel_stage", "fused_x_stage"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_emulated_kernels_match_oracle(case, fused):
    nens, nx, ny, nz, tr, zint, kw, mode_a, seg = CASES[case]
    consts = kw.get("consts", idz.CONSTS_DEFAULT)
    names, pos, mass, idwv = idz.tracer_flags(tr)
    xlen = nx * 500.0
    ylen = ny * 500.0 if ny > 1 else xlen
    f = _inputs(nens, nx, ny, nz, tr, zint, kw, consts, xlen, ylen)
    dz = np.diff(zint)[:, None] * np.ones((1, nens))
    if kw.get("per_ens"):
        dz = dz * (1 + 0.01 * np.arange(nens))[None, :]
    f1, f2 = copy.deepcopy(f), copy.deepcopy(f)
    o = ao.OracleDycore(nens, nx, ny, nz, xlen, ylen, dz, pos, mass, idwv, consts=consts)
    g = eh.EmuDycore(nens, nx, ny, nz, xlen, ylen, dz, pos, mass, idwv, consts=consts, seg=seg)
    o.set_grav_balance(mode_a)
    g.set_grav_balance(mode_a)
    g.set_fused(fused)
    assert g.vz_per_ens == bool(kw.get("per_ens"))
    o.declare_current_profile_as_hydrostatic(f1)
    g.declare_current_profile_as_hydrostatic(f2)
    key = "variable_gravity" if mode_a else "hy_dens_cells"
    assert _rel(g.buffer(key, (nz, nens)), o.variable_gravity if mode_a else o.hy_dens_cells) < 1e-13
    assert g.compute_time_step(f2) == o.compute_time_step(f1)
    for _ in range(2):
        n1, d1 = o.time_step(f1, 2.0)
        n2, d2 = g.time_step(f2, 2.0)
        assert (n1, d1) == (n2, d2)
    assert _rel(f2["density_dry"], f1["density_dry"]) < 1e-13
    assert _rel(f2["temp"], f1["temp"]) < 1e-13
    assert _rel(f2["uvel"], f1["uvel"]) < 1e-11
    assert _rel(f2["wvel"], f1["wvel"]) < 1e-10
    assert _rel(f2["vvel"], f1["vvel"]) < 1e-9
    for t in range(len(tr)):
        assert _rel(f2["tracers"][t], f1["tracers"][t]) < 1e-11
    if kw.get("dry_air"):      # the last stage limited vapour: multipliers < 1 exist (unflagged rows are NaN-poisoned when fused)
        m = g.buffer("mult", (len(tr), nz, ny, nx, nens))
        assert np.nansum(m < 1.0) > 0 and (f1["tracers"] == 0.0).any()


@pytest.mark.parametrize("span,split", [(0, False), (3, False), (0, True), (3, True)],
                         ids=["whole_lines", "spans_of_3", "whole_lines_xtr_launch", "spans_of_3_xtr_launch"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_fused_stage_equals_three_kernel_stage_bit_for_bit(case, span, split):
    """flux(y,z) -> fused x-sweep + state update -> FCT -> tracer update + pressure must reproduce flux(x,y,z) -> FCT ->
    update exactly: same helpers, same rounding points (awfl_device.h: acoustic_face, flux_divergence, rk_combine...).
    Odd and even numbers of sub-steps exercise both parities of the three-buffer rotation."""
    nens, nx, ny, nz, tr, zint, kw, mode_a, seg = CASES[case]
    consts = kw.get("consts", idz.CONSTS_DEFAULT)
    names, pos, mass, idwv = idz.tracer_flags(tr)
    xlen = nx * 500.0
    ylen = ny * 500.0 if ny > 1 else xlen
    f = _inputs(nens, nx, ny, nz, tr, zint, kw, consts, xlen, ylen)
    dz = np.diff(zint)[:, None] * np.ones((1, nens))
    if kw.get("per_ens"):
        dz = dz * (1 + 0.01 * np.arange(nens))[None, :]
    out = []
    # three-kernel stage; fused stage; fused stage with the y differences of the state folded into the z sweep's output (3-D: the z
    # sweep stores yz_divergence(), the x-sweep loads that one field -- the schedule of large 3-D ensembles on the device)
    for fused, fold in ((False, False), (True, False), (True, True)):
        ff = copy.deepcopy(f)
        g = eh.EmuDycore(nens, nx, ny, nz, xlen, ylen, dz, pos, mass, idwv, consts=consts, seg=seg)
        g.set_grav_balance(mode_a)
        g.set_fused(fused)
        if fused:
            g.set_span(span)      # the fused x-sweep cut into spans (each recomputes its closing face) vs whole-line three-kernel stage
            g.set_xtr_split(split)  # tracers 1.. finished inline after the state pass, or in a launch of their own (awfl_xtr_kernel)
            g.set_yz_fold(fold)
        g.declare_current_profile_as_hydrostatic(ff)
        ncyc = [g.time_step(ff, dt)[0] for dt in (2.0, 0.7, 2.0)]
        out.append((ncyc, ff))
    assert out[0][0] == out[1][0] == out[2][0] and any(n % 2 for n in out[0][0]) and any(n % 2 == 0 for n in out[0][0]), out[0][0]
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
        assert np.array_equal(out[0][1][k], out[1][1][k]), k
        assert np.array_equal(out[0][1][k], out[2][1][k]), k + " (folded)"


def test_vertical_tables_match_oracle_matrices():
    """Dycore.h:904-937 restated twice (oracle C, product C++): the DataManager entries must agree exactly."""
    nens, nz = 3, 9
    zint = idz.stretched_interfaces(nz, 12000.0)
    dz = np.diff(zint)[:, None] * (1 + 0.02 * np.arange(nens))[None, :]
    names, pos, mass, idwv = idz.tracer_flags(idz.TRACERS_NONE)
    o = ao.OracleDycore(nens, 6, 1, nz, 3000.0, 3000.0, dz, pos, mass, idwv)
    g = eh.EmuDycore(nens, 6, 1, nz, 3000.0, 3000.0, dz, pos, mass, idwv)
    assert np.array_equal(g.buffer("vert_sten_to_coefs", (nz + 2, 5, 5, nens)), o.vert_sten_to_coefs)
    assert np.array_equal(g.buffer("vert_weno_recon_lower", (nz + 2, 3, 3, 3, nens)), o.vert_weno_recon_lower)
