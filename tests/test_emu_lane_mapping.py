"""CPU-only: the lane mappings of small ensembles in the host emulation of the kernel bodies (tests/emu/, test infrastructure).

Flat lanes (the y/z sweeps take items of the flattened (x, member) axis: flat_lane) and tile kernels (the x direction with a lane
per cell: xtile_* in pam_amd/csrc/awfl_device.h) must reproduce the member-lane sweeps BIT FOR BIT: the same helpers on the same
five stencil values of every cell, only the lane that computes them differs.  Also against the oracle (tolerances as
tests/test_emu_parity.py), and the tile geometry rules on their own.  The device build of the same bodies: tests/test_lane_mapping.py
(-m gpu)."""
import copy

import numpy as np
import pytest

from oracle import awfl_oracle as ao
from pam_amd import idealized as idz
import emu_harness as eh
from test_emu_parity import _inputs, _rel

# name: (nens, nx, ny, nz, tracers, zint, kw, mode_a, tile overrides (W, tc, lpb) to try besides the automatic geometry)
CASES = {
    "c1_like_nens2_3d": (2, 8, 5, 9, idz.TRACERS_NONE, idz.stretched_interfaces(9, 12000.0), {}, True, [(0, 3, 0), (1, 0, 0)]),
    "nens1_3d_nt4_B": (1, 7, 4, 8, idz.TRACERS_KESSLER_SHOC, idz.stretched_interfaces(8, 12000.0), {}, False, [(0, 2, 0), (0, 0, 3)]),
    "nens1_2d_nt10_p3": (1, 9, 1, 8, idz.TRACERS_P3_SHOC, idz.stretched_interfaces(8, 12000.0), dict(consts=idz.CONSTS_P3), True,
                         [(0, 4, 0)]),
    "nens3_perens_nt4": (3, 6, 4, 8, idz.TRACERS_KESSLER_SHOC, idz.stretched_interfaces(8, 12000.0), dict(per_ens=True), True,
                         [(2, 0, 0), (2, 4, 0)]),
    "nens5_vapour_limited_A": (5, 8, 5, 9, idz.TRACERS_NONE, idz.stretched_interfaces(9, 12000.0), dict(dry_air=True), True,
                               [(0, 5, 0), (3, 3, 0)]),
    "nens4_vapour_limited_2d_B": (4, 10, 1, 8, idz.TRACERS_NONE, idz.stretched_interfaces(8, 12000.0), dict(dry_air=True), False,
                                  [(0, 1, 0)]),
}


def _setup(case):
    nens, nx, ny, nz, tr, zint, kw, mode_a, tiles = CASES[case]
    consts = kw.get("consts", idz.CONSTS_DEFAULT)
    names, pos, mass, idwv = idz.tracer_flags(tr)
    xlen = nx * 500.0
    ylen = ny * 500.0 if ny > 1 else xlen
    f = _inputs(nens, nx, ny, nz, tr, zint, kw, consts, xlen, ylen)
    dz = np.diff(zint)[:, None] * np.ones((1, nens))
    if kw.get("per_ens"):
        dz = dz * (1 + 0.01 * np.arange(nens))[None, :]
    return (nens, nx, ny, nz, xlen, ylen, dz, pos, mass, idwv, consts, mode_a), f, tiles


def _run_emu(args, f, flat, xtile, tile=(0, 0, 0), span=0, ftile=None):
    nens, nx, ny, nz, xlen, ylen, dz, pos, mass, idwv, consts, mode_a = args
    ff = copy.deepcopy(f)
    g = eh.EmuDycore(nens, nx, ny, nz, xlen, ylen, dz, pos, mass, idwv, consts=consts)
    g.set_grav_balance(mode_a)
    g.set_fused(True)
    g.set_span(span)
    g.set_lane_mapping(flat, xtile)
    g.set_x_tile(*tile)
    if ftile is not None:
        g.set_flux_tile(True, *ftile)
        g.set_tile_pressure(xtile)         # (with the y/z tile kernels also: the pressure pass inside the x tile kernel)
    g.declare_current_profile_as_hydrostatic(ff)
    ncyc = [g.time_step(ff, dt)[0] for dt in (2.0, 0.7)]
    return ncyc, ff, g


@pytest.mark.parametrize("case", sorted(CASES))
def test_flat_lanes_and_tile_kernels_equal_member_lane_sweeps_bit_for_bit(case):
    args, f, tiles = _setup(case)
    n0, ref, _ = _run_emu(args, f, False, False)
    variants = [("flat y/z lanes", True, False, (0, 0, 0), None), ("tile x kernels", False, True, (0, 0, 0), None),
                ("flat + tile", True, True, (0, 0, 0), None)] + [("flat + tile %r" % (t,), True, True, t, None) for t in tiles]
    # the y/z fluxes as tile kernels too: automatic tiles, short tiles (several per line / column, halo rows), whole columns
    variants += [("y/z tiles", True, True, (0, 0, 0), (0, 0)), ("y/z tiles of 2 / 3", True, False, (0, 0, 0), (2, 3)),
                 ("y/z tiles of 3 / 14", True, True, (0, 0, 0), (3, 14))]
    for name, flat, xtile, tile, ftile in variants:
        n1, got, g = _run_emu(args, f, flat, xtile, tile, ftile=ftile)
        assert n0 == n1, name
        for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
            assert np.isfinite(got[k]).all(), (name, k)
            assert np.array_equal(ref[k], got[k]), (name, k, g.x_tile_geometry(), np.abs(ref[k] - got[k]).max())


@pytest.mark.parametrize("case", ["c1_like_nens2_3d", "nens1_3d_nt4_B", "nens5_vapour_limited_A"])
def test_flat_and_tile_mapping_matches_oracle(case):
    args, f, _ = _setup(case)
    nens, nx, ny, nz, xlen, ylen, dz, pos, mass, idwv, consts, mode_a = args
    f1 = copy.deepcopy(f)
    o = ao.OracleDycore(nens, nx, ny, nz, xlen, ylen, dz, pos, mass, idwv, consts=consts)
    o.set_grav_balance(mode_a)
    o.declare_current_profile_as_hydrostatic(f1)
    n1 = [o.time_step(f1, dt)[0] for dt in (2.0, 0.7)]
    n2, f2, _ = _run_emu(args, f, True, True)
    assert n1 == n2
    assert _rel(f2["density_dry"], f1["density_dry"]) < 1e-13
    assert _rel(f2["temp"], f1["temp"]) < 1e-13
    assert _rel(f2["uvel"], f1["uvel"]) < 1e-11
    assert _rel(f2["wvel"], f1["wvel"]) < 1e-10
    for t in range(f1["tracers"].shape[0]):
        assert _rel(f2["tracers"][t], f1["tracers"][t]) < 1e-11


def test_tile_geometry_rules():
    """whole periodic lines when they fit a workgroup (several short ones per workgroup), else even tiles with one halo row per side"""
    names, pos, mass, idwv = idz.tracer_flags(idz.TRACERS_NONE)

    def geom(nens, nx, tile=(0, 0, 0)):
        nz = 4
        g = eh.EmuDycore(nens, nx, 1, nz, 1000.0 * nx, 1000.0 * nx, np.full((nz, nens), 100.0), pos, mass, idwv)
        g.set_x_tile(*tile)
        return g.x_tile_geometry()
    def used(g, nx):
        t = g["W"] * (g["tc"] + 2 * g["halo"]) * g["lpb"]
        return t / (64.0 * ((t + 63) // 64)), t
    g = geom(1, 32)                       # C2 grid, one member: several lines of 32 lanes per workgroup, full wavefronts
    assert (g["W"], g["nmb"], g["tc"], g["halo"], g["ntl"]) == (1, 1, 32, 0, 1) and used(g, 32)[0] == 1.0 and used(g, 32)[1] >= 192
    g = geom(1, 250)                      # the reference's input shape: a line of 250 lanes per workgroup
    assert (g["halo"], g["tc"], g["lpb"], g["W"]) == (0, 250, 1, 1)
    g = geom(1, 65)                       # the reference's CI shape (inputs/ci/input_pama.yaml): 65 lanes per line
    assert g["halo"] == 0 and used(g, 65)[0] >= 0.9
    g = geom(2, 32)                       # C1: 64 lanes per line
    assert (g["W"], g["halo"]) == (2, 0) and used(g, 32)[0] == 1.0
    g = geom(32, 32)                      # 1024 lanes: rows of 16 members, whole lines of 512 lanes (two workgroups per CU)
    assert (g["W"], g["nmb"], g["halo"], g["tc"], g["lpb"]) == (16, 2, 0, 32, 1)
    g = geom(24, 32)                      # 768 lanes: rows of 12 members would be 96-byte runs -- all 24 members per row
    assert (g["W"], g["halo"], g["tc"]) == (24, 0, 32)
    g = geom(63, 32)                      # 2016 lanes: rows of a quarter of the members (16), whole lines of 512 lanes
    assert (g["W"], g["nmb"], g["halo"], g["tc"]) == (16, 4, 0, 32)
    g = geom(8, 250)                      # 2000 lanes, but rows of 4 members would be 32-byte runs: all 8 members per row, tiles with halo rows
    assert g["W"] == 8 and g["halo"] == 1 and (g["tc"] + 2) * 8 <= 1024 and g["ntl"] * g["tc"] >= 250
    g = geom(128, 32)                     # member blocks of 64
    assert (g["W"], g["nmb"], g["halo"]) == (64, 2, 1) and (g["tc"] + 2) * 64 <= 1024
    g = geom(128, 32, (32, 0, 0))         # rows of 32 members: a whole line of 32 cells fits
    assert (g["W"], g["nmb"], g["halo"], g["tc"]) == (32, 4, 0, 32)


def _random_shapes(n, seed):
    rng = np.random.RandomState(seed)
    shapes = []
    while len(shapes) < n:
        nens = int(rng.choice([1, 2, 3, 5, 7, 17, 33, 40]))
        ny = int(rng.choice([1, 1, 3, 4]))
        nx = int(rng.choice([3, 4, 5, 9, 16, 33]))
        nz = int(rng.choice([3, 5, 8]))
        nt = int(rng.choice([1, 2, 4]))
        if nens * nx * ny * nz * (6 + nt) > 60000:
            continue
        shapes.append((nens, nx, ny, nz, nt, bool(rng.randint(2)), bool(rng.randint(2))))
    return shapes


@pytest.mark.parametrize("shape", _random_shapes(8, 4), ids=lambda s: "nens%d_%dx%dx%d_nt%d_%s%s" % (s[0], s[1], s[2], s[3], s[4], "A" if s[5] else "B", "_limited" if s[6] else ""))
def test_random_shapes_every_mapping_equals_member_lane_sweeps_bit_for_bit(shape):
    """the index logic of the flat lanes and of every tile kernel (x, y/z, fused small-ensemble form) on shapes nobody picked; the same
    test on the device build: tests/test_lane_mapping.py"""
    nens, nx, ny, nz, nt, mode_a, limited = shape
    tr = [("water_vapor", True, True)] + [("t%d" % i, i % 3 != 1, i % 2 == 0) for i in range(nt - 1)]
    if nt >= 3:
        tr = tr[1:3] + tr[:1] + tr[3:]
    names, pos, mass, idwv = idz.tracer_flags(tr)
    zint = idz.stretched_interfaces(nz, 9000.0)
    xlen, ylen = nx * 500.0, (ny if ny > 1 else nx) * 500.0
    f = idz.supercell_fields(nens, nx, ny, nz, zint, tracers=tr, magnitude=0.5)
    idz.add_tracer_blobs(f, tr, xlen, ylen, zint)
    if limited:
        f["uvel"] -= 25.0
        idz.carve_dry_air(f, tr)
    dz = np.diff(zint)[:, None] * np.ones((1, nens))
    args = (nens, nx, ny, nz, xlen, ylen, dz, pos, mass, idwv, idz.CONSTS_DEFAULT, mode_a)
    n0, ref, _ = _run_emu(args, f, False, False)
    for flat, xtile, ftile in ((True, True, None), (True, True, (0, 0)), (True, True, (2, 3))):
        n1, got, g = _run_emu(args, f, flat, xtile, ftile=ftile)
        assert n0 == n1
        for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
            assert np.isfinite(got[k]).all(), (ftile, k)
            assert np.array_equal(ref[k], got[k]), (ftile, k, g.x_tile_geometry(), np.abs(ref[k] - got[k]).max())
