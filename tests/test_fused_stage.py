"""GPU: the fused stage (flux(y,z) -> x-sweep + state update in one kernel -> FCT -> tracer update + pressure; the default)
against the three-kernel stage (flux(x,y,z) -> FCT -> update) of the same library: bit for bit, because both are built from
the same helpers with explicit rounding points (awfl_device.h).  Parity of either against the oracle is
tests/test_gpu_parity.py (which runs the default, i.e. the fused stage)."""
import copy

import numpy as np
import pytest

from pam_amd import idealized as idz

pytestmark = pytest.mark.gpu

CASES = {
    # name: (nens, nx, ny, nz, tracers, zint, per_ens, mode_a, consts)
    "3d_nt1_L60grid": (66, 32, 4, 60, idz.TRACERS_NONE, idz.l60_interfaces(), False, True, idz.CONSTS_DEFAULT),
    "3d_nt4_stretched_B": (3, 6, 6, 8, idz.TRACERS_KESSLER_SHOC, idz.stretched_interfaces(8, 12000.0), False, False, idz.CONSTS_DEFAULT),
    "2d_nt10_perens_p3": (70, 32, 1, 12, idz.TRACERS_P3_SHOC, idz.stretched_interfaces(12, 12000.0), True, True, idz.CONSTS_P3),
    "3d_minimal_3x3x3": (1, 3, 3, 3, idz.TRACERS_NONE, idz.uniform_interfaces(3, 3000.0), False, True, idz.CONSTS_DEFAULT),
    # >= 4096 (line, member block) wavefronts: the fused x-sweep keeps the tracer sweeps inline (smaller cases launch them apart)
    "3d_nt4_many_lines_inline_tracers": (140, 8, 32, 60, idz.TRACERS_KESSLER_SHOC, idz.l60_interfaces(), False, True, idz.CONSTS_DEFAULT),
    "3d_nx64_widest_fused_line": (2, 64, 3, 5, idz.TRACERS_NONE, idz.uniform_interfaces(5, 5000.0), False, True, idz.CONSTS_DEFAULT),
    # member ranges aligned to 64: every wavefront of the pointwise kernels is one row of the FCT flags, and the multipliers
    # of rows without a limited member are not even stored (FctRows, awfl_device.h); members differ (see _run), so flagged
    # rows hold limited and unlimited members side by side
    "3d_nt4_whole_flag_rows": (128, 6, 6, 10, idz.TRACERS_KESSLER_SHOC, idz.stretched_interfaces(10, 12000.0), False, True, idz.CONSTS_DEFAULT),
    # water_vapor as the only tracer, limited (exact zeros beside moist air): the NT=1 tail (awfl_ptail_kernel + awfl_trfix_kernel)
    # against the three-kernel stage, whole flag rows (128 = 2 x 64) and a ragged member count
    "3d_nt1_vapour_limited_rows": (128, 6, 5, 9, idz.TRACERS_NONE, idz.stretched_interfaces(9, 12000.0), False, True, idz.CONSTS_DEFAULT),
    "3d_nt1_vapour_limited_ragged_B": (70, 6, 5, 9, idz.TRACERS_NONE, idz.stretched_interfaces(9, 12000.0), False, False, idz.CONSTS_DEFAULT),
    "2d_nt10_whole_flag_rows": (192, 32, 1, 12, idz.TRACERS_P3_SHOC, idz.stretched_interfaces(12, 12000.0), False, False, idz.CONSTS_P3),
    # per-member vertical grids with MEMBER lanes (awfl_fluxz_pe_kernel: the levels' tables staged in LDS per workgroup of four
    # columns): whole blocks of 64 members + two ranges; a ragged last block and a column count that is not a multiple of four
    # (clamped lanes / wavefronts redo the last member / column); 2-D with many tracers (the pairs in a launch of their own)
    "3d_nt4_perens_member_lanes": (128, 6, 5, 9, idz.TRACERS_KESSLER_SHOC, idz.stretched_interfaces(9, 12000.0), "mod16", True, idz.CONSTS_DEFAULT),
    "3d_nt1_perens_ragged_130_B": (130, 5, 3, 7, idz.TRACERS_NONE, idz.stretched_interfaces(7, 9000.0), "mod16", False, idz.CONSTS_DEFAULT),
    "2d_nt10_perens_member_lanes": (192, 32, 1, 12, idz.TRACERS_P3_SHOC, idz.stretched_interfaces(12, 12000.0), "mod16", True, idz.CONSTS_P3),
}


def _run(case, fused, chunks=0, want_mult=False, fold=None, lanes=None, tail=None):
    import torch
    from pam_amd import Dycore, PamCoupler
    nens, nx, ny, nz, tr, zint, per_ens, mode_a, consts = CASES[case]
    xlen = nx * 500.0
    ylen = ny * 500.0 if ny > 1 else xlen
    f = idz.supercell_fields(nens, nx, ny, nz, zint, consts=consts, tracers=tr, magnitude=0.5)
    idz.add_tracer_blobs(f, tr, xlen, ylen, zint)
    if "vapour_limited" in case:
        f["uvel"] -= 25.0
        f["vvel"] += 7.0
        idz.carve_dry_air(f, tr)
    if nens >= 64:      # every third member: blobs on a positive floor (the limiter stays idle there); every fifth: none at all
        for t, (name, _, _) in enumerate(tr):
            if name == "water_vapor":
                continue
            q = f["tracers"][t]
            q[..., 1::3] = q[..., 1::3] + 2.0e-4 * f["density_dry"][..., 1::3]
            q[..., 2::5] = 0.0
    zi = np.asarray(zint)[:, None] * np.ones((1, nens))
    if per_ens == "mod16":
        zi = zi * (1 + 0.01 * (np.arange(nens) % 16) + 1.0e-4 * (np.arange(nens) // 16))[None, :]
    elif per_ens:
        zi = zi * (1 + 0.01 * np.arange(nens))[None, :]
    coupler = PamCoupler("cuda:0")
    coupler.set_option("crm_dt", 2.0)
    for k, v in consts.items():
        coupler.set_option(k, v)
    coupler.allocate_coupler_state(nz, ny, nx, nens)
    coupler.set_grid(xlen, ylen, zi)
    for n, p, m in tr:
        coupler.add_tracer(n, "", p, m)
    dycore = Dycore()
    dycore.init(coupler)
    dycore.set_fused_stage(fused)
    if chunks:
        dycore.set_ensemble_chunks(chunks)
    if lanes is not None:
        dycore.set_lane_mapping(*lanes)
    if fold is not None:
        dycore.set_yz_fold(fold)
    if tail is not None:
        dycore.set_tail_fusion(tail)
    coupler.load_fields(f)
    if not mode_a:
        coupler.set_option("balance_hydrostasis_with_gravity", False)
    dycore.declare_current_profile_as_hydrostatic(coupler)
    ncyc = []
    for crm_dt in (2.0, 0.7, 2.0):        # odd and even sub-step counts: both parities of the three-buffer rotation
        coupler.set_option("crm_dt", crm_dt)
        ncyc.append(dycore.timeStep(coupler))
    torch.cuda.synchronize()
    out = coupler.dump_fields()
    if want_mult:       # FCT multipliers of the last stage (complete only in the three-kernel stage)
        out["mult"] = dycore.debug_buffer("mult").cpu().numpy().reshape(len(tr), nz, ny, nx, nens)
    dycore.finalize(coupler)
    return ncyc, out


@pytest.mark.parametrize("case", sorted(CASES))
def test_fused_stage_equals_three_kernel_stage_bit_for_bit(case):
    n0, a = _run(case, fused=False)
    n1, b = _run(case, fused=True)
    assert n0 == n1
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
        assert np.isfinite(a[k]).all(), k
        assert np.array_equal(a[k], b[k]), (k, np.abs(a[k] - b[k]).max())


@pytest.mark.parametrize("case", ["3d_nt4_whole_flag_rows", "2d_nt10_whole_flag_rows", "3d_nt1_vapour_limited_rows"])
def test_flag_row_cases_exercise_the_limiter(case):
    """the cases above are only worth something if rows ARE flagged, and if flagged rows mix limited and unlimited members"""
    _, a = _run(case, fused=False, want_mult=True)
    m = a["mult"]
    nt, nz, ny, nx, nens = m.shape
    assert np.isfinite(m).all() and (m < 1.0).any() and (m == 1.0).any()
    rows = m.reshape(nt, nz, ny, nx, nens // 64, 64)
    lim = (rows < 1.0).sum(axis=-1)
    assert ((lim > 0) & (lim < 64)).any(), "no row holds limited and unlimited members side by side"
    assert (lim == 0).any(), "rows without any limited member must exist too (they are the ones that are skipped)"


@pytest.mark.parametrize("case", ["3d_nt4_whole_flag_rows", "3d_nt1_vapour_limited_rows", "3d_nt4_perens_member_lanes", "3d_nt1_perens_ragged_130_B",
                                  "3d_nt4_many_lines_inline_tracers"])
def test_yz_fold_is_a_schedule_not_an_arithmetic(case):
    """3-D member lanes: the z sweep storing the y+z part of the state's divergence (one field per variable for the x-sweep; the
    default) == the z sweep storing its own differences and the x-sweep loading both (pam_amd_awfl_set_yz_fold), bit for bit"""
    n0, a = _run(case, fused=True, fold="off")
    n1, b = _run(case, fused=True, fold="on")
    assert n0 == n1
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
        assert np.isfinite(a[k]).all(), k
        assert np.array_equal(a[k], b[k]), (k, np.abs(a[k] - b[k]).max())


@pytest.mark.parametrize("case", ["3d_nt4_whole_flag_rows", "2d_nt10_whole_flag_rows", "3d_nt4_many_lines_inline_tracers",
                                  "2d_nt10_perens_member_lanes", "3d_nt4_perens_member_lanes"])
@pytest.mark.parametrize("chunks", [1, 2])
def test_tail_fusion_is_a_schedule_not_an_arithmetic(case, chunks):
    """NT > 1, member-lane sweeps: phase 2 of the further tracers + the pressure pass + water vapour's fix-up as ONE launch
    (awfl_xtr2_tail_kernel: three further tracers go one per wavefront, nine in pairs) == three launches, one and two member ranges"""
    n0, a = _run(case, fused=True, tail="off", chunks=chunks)
    n1, b = _run(case, fused=True, tail="on", chunks=chunks)
    assert n0 == n1
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
        assert np.isfinite(a[k]).all(), k
        assert np.array_equal(a[k], b[k]), (k, np.abs(a[k] - b[k]).max())


def test_yz_fold_is_refused_where_it_does_not_exist():
    """2-D grids have no y part; flat lanes / tile kernels form the sum themselves"""
    import torch  # noqa: F401
    from pam_amd import Dycore, PamCoupler
    from pam_amd.capi import PamAmdError
    for nens, ny in ((128, 1), (3, 6)):
        coupler = PamCoupler("cuda:0")
        coupler.set_option("crm_dt", 2.0)
        coupler.allocate_coupler_state(8, ny, 6, nens)
        coupler.set_grid(3000.0, 3000.0, idz.stretched_interfaces(8, 12000.0))
        coupler.add_tracer("water_vapor", "", True, True)
        dycore = Dycore()
        dycore.init(coupler)
        with pytest.raises(PamAmdError):
            dycore.set_yz_fold("on")
        dycore.set_yz_fold("off")
        dycore.set_yz_fold("auto")
        dycore.finalize(coupler)


@pytest.mark.parametrize("case", ["3d_nt4_perens_member_lanes", "3d_nt1_perens_ragged_130_B", "2d_nt10_perens_member_lanes"])
def test_per_member_grids_member_lanes_equal_flat_lanes_bit_for_bit(case):
    """per-member vertical grids: the LDS-staged tables of the member-lane z sweep (awfl_fluxz_pe_kernel) against flat lanes, where
    every lane reads its member's table from global memory (ZTabLane) -- the same weno5_table on the same 31 values"""
    n0, a = _run(case, fused=True, lanes=("flat", "sweep"))
    n1, b = _run(case, fused=True, lanes=("member", "sweep"))
    assert n0 == n1
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
        assert np.isfinite(a[k]).all(), k
        assert np.array_equal(a[k], b[k]), (k, np.abs(a[k] - b[k]).max())


def test_fct_flag_rows_are_chunking_invariant():
    """ranges of 64 members each carry their own rows of FCT flags (awfl_kernels.hip: fct_rows) == one range"""
    _, a = _run("3d_nt4_whole_flag_rows", fused=True, chunks=1)
    _, b = _run("3d_nt4_whole_flag_rows", fused=True, chunks=2)
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
        assert np.array_equal(a[k], b[k]), k


def test_fused_nt1_limiter_path_is_chunking_invariant():
    """the NT=1 fix-up pass with two member ranges (each range's wavefronts are whole flag rows) == one range"""
    _, a = _run("3d_nt1_vapour_limited_rows", fused=True, chunks=1)
    _, b = _run("3d_nt1_vapour_limited_rows", fused=True, chunks=2)
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
        assert np.array_equal(a[k], b[k]), k


def test_fused_stage_is_chunking_invariant():
    """two member ranges on internal streams (fork/join, buffer rotation per range) == one range"""
    _, a = _run("3d_nt1_L60grid", fused=True, chunks=1)
    _, b = _run("3d_nt1_L60grid", fused=True, chunks=2)
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
        assert np.array_equal(a[k], b[k]), k


def test_fused_stage_handles_long_lines():
    """no LDS slots any more: a line of any length is swept by its wavefront (nx = 80 here)"""
    import torch
    from pam_amd import Dycore, PamCoupler
    nens, nx, ny, nz = 2, 80, 1, 5
    res = []
    for fused in (False, True):
        coupler = PamCoupler("cuda:0")
        coupler.set_option("crm_dt", 1.0)
        coupler.allocate_coupler_state(nz, ny, nx, nens)
        zint = idz.uniform_interfaces(nz, 5000.0)
        coupler.set_grid(nx * 500.0, nx * 500.0, zint)
        coupler.add_tracer("water_vapor", "", True, True)
        dycore = Dycore()
        dycore.init(coupler)
        dycore.set_fused_stage(fused)
        coupler.load_fields(idz.supercell_fields(nens, nx, ny, nz, zint, magnitude=0.5))
        dycore.declare_current_profile_as_hydrostatic(coupler)
        dycore.timeStep(coupler)
        torch.cuda.synchronize()
        res.append(coupler.dump_fields())
        dycore.finalize(coupler)
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k
