"""GPU: the lane mappings of the fused stage (include/pam_amd_awfl.h: pam_amd_awfl_set_lane_mapping).

Small ensembles (nens < 64: every input file the reference ships has nens = 1, standalone/mmf_simplified/inputs/input_pama.yaml:14)
run with FLAT lanes in the y/z sweeps and TILE kernels in x (a lane per cell, neighbours at lane distance nens through LDS);
large ones with MEMBER lanes and sweeps.  Same helpers, same rounding points: the two must agree BIT FOR BIT
  * on the same input (both mappings forced on small, 64-aligned and ragged ensembles, several tile geometries),
  * on a tiled input (nens = 2 with flat lanes against the same two members tiled to 64 and run with member lanes),
and the small-ensemble defaults must really be flat / tile, with wavefronts >= 90 % full whenever nx*nens >= 64.
Parity of the new path against the ORACLE: tests/test_gpu_parity.py -- its cases with nens < 64 (C1 exactly among them) run it by
default."""
import numpy as np
import pytest

from pam_amd import idealized as idz

pytestmark = pytest.mark.gpu

CASES = {
    # name: (nens, nx, ny, nz, tracers, zint, per_ens, mode_a, consts, limiter on vapour, tile geometries (W, tc, lpb) besides the automatic one)
    "nens1_c2grid_slab": (1, 32, 6, 20, idz.TRACERS_NONE, idz.stretched_interfaces(20, 15000.0), False, True, idz.CONSTS_DEFAULT, False, [(0, 5, 0), (0, 0, 1)]),
    "nens1_ref_shape_nt4": (1, 250, 1, 16, idz.TRACERS_KESSLER_SHOC, idz.uniform_interfaces(16, 16000.0), False, True, idz.CONSTS_DEFAULT, False, [(0, 100, 0)]),
    "nens2_c1_like": (2, 32, 8, 12, idz.TRACERS_NONE, idz.uniform_interfaces(12, 12000.0), False, True, idz.CONSTS_DEFAULT, False, [(1, 0, 0)]),
    "nens8_nt4_B": (8, 12, 6, 10, idz.TRACERS_KESSLER_SHOC, idz.stretched_interfaces(10, 12000.0), False, False, idz.CONSTS_DEFAULT, False, [(0, 4, 0), (4, 0, 0)]),
    "nens32_nt10_2d_p3_perens": (32, 32, 1, 12, idz.TRACERS_P3_SHOC, idz.stretched_interfaces(12, 12000.0), True, True, idz.CONSTS_P3, False, [(0, 7, 0)]),
    # a line longer than a workgroup (1500 lanes): tiles of cells with halo rows, partial last tile
    "nens1_long_line_halo_tiles": (1, 1500, 1, 6, idz.TRACERS_KESSLER_SHOC, idz.uniform_interfaces(6, 6000.0), False, True, idz.CONSTS_DEFAULT, False, [(0, 700, 0)]),
    "nens5_vapour_limited": (5, 16, 5, 9, idz.TRACERS_NONE, idz.stretched_interfaces(9, 12000.0), False, True, idz.CONSTS_DEFAULT, True, [(0, 3, 0)]),
    "nens40_vapour_limited_B": (40, 32, 3, 9, idz.TRACERS_NONE, idz.stretched_interfaces(9, 12000.0), False, False, idz.CONSTS_DEFAULT, True, [(0, 6, 0)]),
    # large ensembles with the small-ensemble mappings forced: 64-aligned (a wavefront of a 64-lane row IS a row of FCT flags: sparse
    # multiplier stores), ragged, rows of 32 members (a whole 32-cell line in one workgroup)
    "nens128_nt4_rows": (128, 32, 4, 10, idz.TRACERS_KESSLER_SHOC, idz.stretched_interfaces(10, 12000.0), False, True, idz.CONSTS_DEFAULT, False, [(32, 0, 0), (64, 6, 0)]),
    "nens70_ragged_vapour_limited": (70, 12, 5, 9, idz.TRACERS_NONE, idz.stretched_interfaces(9, 12000.0), False, True, idz.CONSTS_DEFAULT, True, [(35, 0, 0)]),
    "nens192_nt10_2d": (192, 32, 1, 12, idz.TRACERS_P3_SHOC, idz.stretched_interfaces(12, 12000.0), False, False, idz.CONSTS_P3, False, [(64, 14, 0)]),
    # whole lines inside ONE wavefront (nx x row lanes divides 64): the x tile kernels exchange by wavefront shuffles (SHUFFLE_CASES below)
    "nens1_nx16_nt4": (1, 16, 4, 10, idz.TRACERS_KESSLER_SHOC, idz.stretched_interfaces(10, 12000.0), False, True, idz.CONSTS_DEFAULT, False, [(0, 0, 1), (0, 0, 3)]),
    "nens4_nx16_nt10_2d_p3_B": (4, 16, 1, 12, idz.TRACERS_P3_SHOC, idz.stretched_interfaces(12, 12000.0), True, False, idz.CONSTS_P3, False, [(2, 0, 0)]),
    "nens2_nx32_vapour_limited": (2, 32, 5, 9, idz.TRACERS_NONE, idz.stretched_interfaces(9, 12000.0), False, True, idz.CONSTS_DEFAULT, True, [(1, 0, 0)]),
}
SHUFFLE_CASES = ("nens1_c2grid_slab", "nens2_c1_like", "nens1_nx16_nt4", "nens4_nx16_nt10_2d_p3_B", "nens2_nx32_vapour_limited")


def _fields(case):
    nens, nx, ny, nz, tr, zint, per_ens, mode_a, consts, limiter, tiles = CASES[case]
    xlen = nx * 500.0
    ylen = ny * 500.0 if ny > 1 else xlen
    f = idz.supercell_fields(nens, nx, ny, nz, zint, consts=consts, tracers=tr, magnitude=0.5)
    idz.add_tracer_blobs(f, tr, xlen, ylen, zint)
    if limiter:
        f["uvel"] -= 25.0
        f["vvel"] += 7.0
        idz.carve_dry_air(f, tr)
    return f, xlen, ylen


def _run(case, f, xlen, ylen, yz, xk, tile=(0, 0, 0), nens_override=None, ftile=("auto", 0, 0), graph="auto", dts=(2.0, 0.7), xex="auto",
         fusion=None):
    import torch
    from pam_amd import Dycore, PamCoupler
    nens, nx, ny, nz, tr, zint, per_ens, mode_a, consts, limiter, tiles = CASES[case]
    if nens_override:
        nens = nens_override
    zi = np.asarray(zint)[:, None] * np.ones((1, nens))
    if per_ens:
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
    dycore.set_lane_mapping(yz, xk)
    dycore.set_x_tile(*tile)
    if xex != "auto":
        dycore.set_x_exchange(xex)
    dycore.set_flux_tile(*ftile)
    if fusion:
        dycore.set_flux_tile_parts("beside" if fusion == "beside" else "behind")
        dycore.set_tile_state_parts("parts" if fusion == "beside" else "one")
    dycore.set_tile_fusion(fusion or ("inside" if ftile[0] == "tile" else ("separate" if ftile[0] == "sweep" else "auto")))
    dycore.set_graph_replay(graph)
    mapping = dycore.get_lane_mapping()
    coupler.load_fields(f)
    if not mode_a:
        coupler.set_option("balance_hydrostasis_with_gravity", False)
    dycore.declare_current_profile_as_hydrostatic(coupler)
    ncyc = []
    for crm_dt in dts:
        coupler.set_option("crm_dt", crm_dt)
        ncyc.append(dycore.timeStep(coupler))
    torch.cuda.synchronize()
    rows = dycore.debug_fct_rows()
    out = coupler.dump_fields()
    dycore.finalize(coupler)
    return ncyc, out, mapping, rows


@pytest.mark.parametrize("case", sorted(CASES))
def test_flat_lanes_and_tile_kernels_equal_member_lanes_and_sweeps_bit_for_bit(case):
    f, xlen, ylen = _fields(case)
    tiles = CASES[case][-1]
    n0, ref, m0, rows = _run(case, f, xlen, ylen, "member", "sweep")
    assert not m0["yz_flat"] and not m0["x_tiles"]
    if CASES[case][-2]:
        assert 0 < rows[0] <= rows[1] and rows[2], rows     # the limiter acted on vapour in the last stage: the sparse paths ran
    # (y/z lanes, x kernels, x tile geometry, (y/z fluxes as ONE tile kernel, cells per y tile, levels per z tile))
    variants = [("flat", "sweep", (0, 0, 0), ("tile", 0, 0)), ("member", "tile", (0, 0, 0), ("auto", 0, 0)), ("flat", "tile", (0, 0, 0), ("tile", 0, 0)),
                ("flat", "tile", (0, 0, 0), ("sweep", 0, 0)), ("flat", "sweep", (0, 0, 0), ("tile", 2, 3)), ("flat", "tile", (0, 0, 0), ("tile", 5, 14))]
    variants += [("flat", "tile", t, ("auto", 0, 0)) for t in tiles]
    variants = [v + (None,) for v in variants]
    # tracer phase 1 in workgroups BESIDE the state pass (they rebuild the face mass flux themselves) and BEHIND it, LDS exchange
    variants += [("flat", "tile", (0, 0, 0), ("auto", 0, 0), "beside"), ("flat", "tile", (0, 0, 0), ("auto", 0, 0), "inside")]
    variants += [("flat", "tile", t, ("auto", 0, 0), "beside") for t in tiles]
    for yz, xk, tile, ftile, fusion in variants:
        n1, got, m1, _ = _run(case, f, xlen, ylen, yz, xk, tile, ftile=ftile, fusion=fusion, xex="lds" if fusion else "auto")
        assert m1["yz_flat"] == (yz == "flat") and m1["x_tiles"] == (xk == "tile"), m1
        if yz == "flat" and ftile[0] != "auto":
            assert m1["yz_tile_kernel"] == (ftile[0] == "tile"), m1
        assert n0 == n1
        for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
            assert np.isfinite(got[k]).all(), (yz, xk, tile, ftile, k)
            assert np.array_equal(ref[k], got[k]), (yz, xk, tile, ftile, m1, k, np.abs(ref[k] - got[k]).max())


@pytest.mark.parametrize("case", SHUFFLE_CASES)
def test_wavefront_shuffle_exchange_equals_the_lds_exchange_and_the_sweeps_bit_for_bit(case):
    """north_star: "wavefront shuffles for the 5-point reconstruction".  Where a whole periodic line of an x tile lies inside one
    wavefront the tile kernels fetch the stencil values, the right-edge values of the cell to the left and the fluxes of the right
    face from the neighbouring LANES (ds_bpermute pairs) -- no LDS image, no workgroup barrier -- and that is the default there.  Same
    values into the same helpers: the same bits as the LDS form and as the member-lane sweeps; state kernel, inline and separately
    launched tracer phases, vapour limited, 2-D and 3-D, mode B, per-member grids, one and several lines per wavefront."""
    f, xlen, ylen = _fields(case)
    tiles = CASES[case][-1]
    n0, ref, m0, rows = _run(case, f, xlen, ylen, "member", "sweep")
    if CASES[case][-2]:
        assert 0 < rows[0] <= rows[1] and rows[2], rows
    for tile in [(0, 0, 0)] + [t for t in tiles if t[1] == 0]:      # (whole-line tiles: tiles of cells with halo rows exchange through LDS)
        for ftile in (("auto", 0, 0), ("sweep", 0, 0)):         # ("sweep": the pressure pass and tracer phase 1 as launches of their own)
            outs = {}
            for xex in ("lds", "shuffle", "auto", "shuffle+beside"):      # (beside: tracer phase 1 in workgroups of its own, shuffle form)
                n1, got, m1, _ = _run(case, f, xlen, ylen, "flat", "tile", tile, ftile=ftile, xex=xex.split("+")[0],
                                      fusion="beside" if "+" in xex else None)
                assert m1["x_tiles"] and m1["x_shuffles"] == (xex != "lds"), (xex, m1)
                assert n1 == n0
                outs[xex] = got
            for xex, got in outs.items():
                for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
                    assert np.isfinite(got[k]).all(), (xex, tile, k)
                    assert np.array_equal(ref[k], got[k]), (xex, tile, ftile, k, np.abs(ref[k] - got[k]).max())


def test_wavefront_shuffles_are_refused_where_a_line_does_not_fit_a_wavefront():
    from pam_amd import PamAmdError
    import torch
    from pam_amd import Dycore, PamCoupler
    case = "nens8_nt4_B"                      # 12-cell lines of 8 members: 96 lanes
    nens, nx, ny, nz, tr, zint = CASES[case][:6]
    coupler = PamCoupler("cuda:0")
    coupler.set_option("crm_dt", 2.0)
    coupler.allocate_coupler_state(nz, ny, nx, nens)
    coupler.set_grid(nx * 500.0, ny * 500.0, zint)
    for n, p, m in tr:
        coupler.add_tracer(n, "", p, m)
    d = Dycore()
    d.init(coupler)
    assert d.get_lane_mapping()["x_tiles"] and not d.get_lane_mapping()["x_shuffles"]
    with pytest.raises(PamAmdError):
        d.set_x_exchange("shuffle")
    d.set_x_exchange("lds")
    d.finalize(coupler)


def test_small_ensemble_defaults_are_flat_and_tile_and_large_ones_member_and_sweep():
    # (flat y/z lanes, x tile kernels): below 64 members both; ragged ensembles below 128 flat y/z lanes with x sweeps; else neither
    for case, want in (("nens1_c2grid_slab", (True, True)), ("nens2_c1_like", (True, True)), ("nens40_vapour_limited_B", (True, True)),
                       ("nens128_nt4_rows", (False, False)), ("nens70_ragged_vapour_limited", (True, False)),
                       ("nens192_nt10_2d", (False, False))):
        f, xlen, ylen = _fields(case)
        _, _, m, _ = _run(case, f, xlen, ylen, "auto", "auto")
        assert (m["yz_flat"], m["x_tiles"]) == want, (case, m)


def test_nens2_flat_lanes_equal_the_same_members_tiled_to_64_member_lanes():
    """the (x, member)-lane path against the member-lane path on a tiled input"""
    case = "nens2_c1_like"
    f, xlen, ylen = _fields(case)
    n0, a, m0, _ = _run(case, f, xlen, ylen, "auto", "auto")
    assert m0["yz_flat"] and m0["x_tiles"]
    f64 = {k: np.ascontiguousarray(np.tile(v, 32)) for k, v in f.items()}     # members 0,1,0,1,...
    n1, b, m1, _ = _run(case, f64, xlen, ylen, "auto", "auto", nens_override=64)
    assert not m1["yz_flat"] and not m1["x_tiles"]
    assert n0 == n1
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
        assert np.array_equal(a[k], b[k][..., :2]) and np.array_equal(a[k], b[k][..., 62:]), k


def _lane_use(nens, nx, ny, nz):
    """fraction of launched lanes that are active in the three kinds of kernels of a small-ensemble stage (from the geometry)"""
    from pam_amd import Dycore, PamCoupler
    zint = idz.uniform_interfaces(nz, 1000.0 * nz)
    coupler = PamCoupler("cuda:0")
    coupler.allocate_coupler_state(nz, ny, nx, nens)
    coupler.set_grid(1000.0 * nx, 1000.0 * max(ny, 1), zint)
    coupler.add_tracer("water_vapor", "", True, True)
    d = Dycore()
    d.init(coupler)
    m = d.get_lane_mapping()
    d.finalize(coupler)
    coupler.dm.finalize()
    assert m["yz_flat"] and m["x_tiles"]
    g = m["tile"]
    t = g["W"] * (g["tc"] + 2 * g["halo"]) * g["lpb"]
    x_use = t / (64.0 * -(-t // 64))
    row = nx * nens
    items_y, items_z = nz * row, ny * row
    y_use = items_y / (64.0 * -(-items_y // 64)) if ny > 1 else 1.0
    z_use = items_z / (64.0 * -(-items_z // 64))
    return x_use, y_use, z_use, g


@pytest.mark.parametrize("nens,nx,ny,nz", [(1, 250, 1, 50), (1, 65, 1, 50), (1, 64, 1, 50), (2, 32, 32, 60), (1, 32, 32, 60), (3, 32, 4, 20),
                                           (8, 32, 32, 60), (32, 32, 32, 60), (33, 32, 1, 60), (63, 32, 1, 60), (7, 100, 3, 10), (1, 96, 1, 60)])
def test_wavefronts_are_at_least_90_percent_full_when_nx_times_nens_reaches_64(nens, nx, ny, nz):
    x_use, y_use, z_use, g = _lane_use(nens, nx, ny, nz)
    # x tiles and y sweeps: rows shorter than a wavefront are packed (lines per workgroup; levels in the flat index space), so they are
    # full whatever nx*nens is.  The z sweep has ny*nx*nens columns in ALL: its only loss is the last, partial wavefront -- below 10 %
    # from 576 columns on; a 2-D CRM of 65 columns (the reference's CI input) is two wavefronts whatever the mapping.
    assert x_use >= 0.90 and y_use >= 0.90, (x_use, y_use, z_use, g)
    cols = ny * nx * nens
    if cols >= 576 or cols % 64 == 0:
        assert z_use >= 0.90, (z_use, cols)
    else:
        assert z_use == cols / (64.0 * -(-cols // 64))


def test_tile_geometry_puts_one_workgroup_on_every_cu_when_the_grid_allows():
    """Small grids are launch-latency problems: the x tile geometry minimises (workgroups per CU, rounded up) x (lanes per workgroup),
    ties going to 256-lane workgroups (round 5: BASELINE's 32x32x60 grid with one member 8 lines per workgroup = 240 workgroups of 256
    lanes instead of 320 of 192; with two members 4 lines = 480 of 256 instead of 640 of 192).  Results do not depend on it (the
    bit-for-bit tests above run both)."""
    import torch
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the expected geometry is that of a 256-CU device")
    for nens, want in ((1, 8), (2, 4), (8, 1)):
        _, _, _, g = _lane_use(nens, 32, 32, 60)
        assert g["lpb"] == want and g["halo"] == 0, (nens, g)


@pytest.mark.parametrize("case", ["nens1_c2grid_slab", "nens1_ref_shape_nt4", "nens8_nt4_B", "nens5_vapour_limited", "nens40_vapour_limited_B"])
def test_time_step_replayed_from_a_hip_graph_equals_eager_launches_bit_for_bit(case):
    """pam_amd_awfl_set_graph_replay: the whole timeStep captured once per (coupler arrays, sub-cycle count, buffer parity) and replayed.
    Seven steps with two crm_dt values: graphs are captured, reused, and both parities of the three-buffer rotation occur (odd and even
    sub-cycle counts); the FCT flag value advances through the device word the replay sets (vapour-limited cases: rows ARE flagged
    and the fix-up works in every stage)."""
    f, xlen, ylen = _fields(case)
    dts = (2.0, 0.7, 2.0, 2.0, 0.7, 0.7, 2.0)
    n0, ref, _, rows0 = _run(case, f, xlen, ylen, "auto", "auto", graph="off", dts=dts)
    n1, got, _, rows1 = _run(case, f, xlen, ylen, "auto", "auto", graph="on", dts=dts)
    assert n0 == n1 and any(n % 2 for n in n0) and any(n % 2 == 0 for n in n0), n0
    assert rows0 == rows1
    if CASES[case][-2]:
        assert rows0[0] > 0
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
        assert np.isfinite(got[k]).all(), k
        assert np.array_equal(ref[k], got[k]), (k, np.abs(ref[k] - got[k]).max())


def _random_shapes(n, seed):
    rng = np.random.RandomState(seed)
    shapes = []
    while len(shapes) < n:
        nens = int(rng.choice([1, 2, 3, 5, 7, 12, 17, 24, 31, 33, 40, 63, 65, 70, 100]))
        ny = int(rng.choice([1, 1, 3, 4, 5, 7]))
        nx = int(rng.choice([3, 4, 5, 9, 16, 33, 50, 67, 130])) if nens <= 12 else int(rng.choice([3, 5, 9, 16, 21, 33]))
        nz = int(rng.choice([3, 5, 8, 13, 17]))
        nt = int(rng.choice([1, 1, 2, 3, 4, 10]))
        if nens * nx * ny * nz * (6 + nt) > 1.2e6:
            continue
        shapes.append((nens, nx, ny, nz, nt, bool(rng.randint(2)), bool(rng.randint(2))))
    return shapes


@pytest.mark.parametrize("shape", _random_shapes(18, 20261004), ids=lambda s: "nens%d_%dx%dx%d_nt%d_%s%s" % (s[0], s[1], s[2], s[3], s[4], "A" if s[5] else "B", "_limited" if s[6] else ""))
def test_random_shapes_automatic_mapping_equals_member_lanes_and_sweeps_bit_for_bit(shape):
    """index logic of every mapping on shapes nobody picked: partial last tiles, ragged member blocks, several short lines per
    workgroup with a partial last group, lines longer than a workgroup, 2-D and 3-D, 1 ... 10 tracers, both hydrostasis modes, vapour
    limited or not -- the automatic mapping of the shape (flat lanes / tile kernels / fused small-ensemble kernel where it applies)
    and the forced flat + tile mapping against member lanes + sweeps"""
    import torch
    from pam_amd import Dycore, PamCoupler
    nens, nx, ny, nz, nt, mode_a, limited = shape
    tr = [("water_vapor", True, True)] + [("t%d" % i, i % 3 != 1, i % 2 == 0) for i in range(nt - 1)]
    if nt >= 3:
        tr = tr[1:3] + tr[:1] + tr[3:]          # water vapour not registered first (P3 registers it last)
    zint = idz.stretched_interfaces(nz, 9000.0)
    xlen, ylen = nx * 500.0, (ny if ny > 1 else nx) * 500.0
    f = idz.supercell_fields(nens, nx, ny, nz, zint, tracers=tr, magnitude=0.5)
    idz.add_tracer_blobs(f, tr, xlen, ylen, zint)
    if limited:
        f["uvel"] -= 25.0
        idz.carve_dry_air(f, tr)

    def run(yz, xk):
        coupler = PamCoupler("cuda:0")
        coupler.set_option("crm_dt", 1.5)
        coupler.allocate_coupler_state(nz, ny, nx, nens)
        coupler.set_grid(xlen, ylen, zint)
        for n, p, m in tr:
            coupler.add_tracer(n, "", p, m)
        d = Dycore()
        d.init(coupler)
        try:
            d.set_lane_mapping(yz, xk)
        except Exception:
            d.finalize(coupler)
            return None
        coupler.load_fields(f)
        if not mode_a:
            coupler.set_option("balance_hydrostasis_with_gravity", False)
        d.declare_current_profile_as_hydrostatic(coupler)
        n = [d.timeStep(coupler) for _ in range(2)]
        torch.cuda.synchronize()
        out = coupler.dump_fields()
        d.finalize(coupler)
        return n, out
    ref = run("member", "sweep")
    for yz, xk in (("auto", "auto"), ("flat", "tile")):
        got = run(yz, xk)
        if got is None:
            continue
        assert got[0] == ref[0]
        for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
            assert np.isfinite(got[1][k]).all(), (yz, xk, k)
            assert np.array_equal(ref[1][k], got[1][k]), (yz, xk, k, np.abs(ref[1][k] - got[1][k]).max())


def test_more_lines_than_the_tile_launch_grid_falls_back_to_sweeps():
    """the groups of x lines are the y dimension of the tile kernels' launch grid (<= 65535): beyond that the automatic mapping keeps
    the x sweeps (flat y/z lanes stay), and the step still equals the member-lane one"""
    import torch
    from pam_amd import Dycore, PamCoupler
    nens, nx, ny, nz = 1, 4, 22000, 3
    tr = idz.TRACERS_NONE
    zint = idz.uniform_interfaces(nz, 3000.0)
    f = idz.supercell_fields(nens, nx, ny, nz, zint, tracers=tr, magnitude=0.5)
    res = []
    for yz, xk in (("auto", "auto"), ("member", "sweep")):
        coupler = PamCoupler("cuda:0")
        coupler.set_option("crm_dt", 1.0)
        coupler.allocate_coupler_state(nz, ny, nx, nens)
        coupler.set_grid(nx * 500.0, ny * 500.0, zint)
        coupler.add_tracer("water_vapor", "", True, True)
        d = Dycore()
        d.init(coupler)
        d.set_lane_mapping(yz, xk)
        if yz == "auto":
            assert d.get_lane_mapping()["x_tiles"]              # 48 lines of 4 lanes per workgroup: 1375 groups
            d.set_x_tile(lines_per_group=1)                     # one line per workgroup: 66000 groups > 65535
        m = d.get_lane_mapping()
        if yz == "auto":
            assert m["yz_flat"] and not m["x_tiles"], m
            with pytest.raises(Exception):
                d.set_lane_mapping("flat", "tile")          # asked for explicitly: refused, nothing changes
        coupler.load_fields(f)
        d.declare_current_profile_as_hydrostatic(coupler)
        d.timeStep(coupler)
        torch.cuda.synchronize()
        res.append(coupler.dump_fields())
        d.finalize(coupler)
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k


@pytest.mark.parametrize("which", [1, 2, 3], ids=["begin_capture", "end_capture", "instantiate"])
def test_a_failed_graph_capture_runs_the_step_eagerly_and_says_so(which):
    """when the capture machinery fails (hipStreamBeginCapture / hipStreamEndCapture / hipGraphInstantiate; injected through
    pam_amd_awfl_debug_fail_next_capture) the step runs eagerly with the runtime's sticky error cleared, the replay is switched off for
    the handle, pam_amd_awfl_last_error() carries a warning, and the results are those of eager launches bit for bit"""
    import torch
    from pam_amd import Dycore, PamCoupler, capi
    nens, nx, ny, nz = 2, 8, 4, 10
    tr = idz.TRACERS_KESSLER_SHOC
    zint = idz.stretched_interfaces(nz, 12000.0)
    xlen, ylen = nx * 500.0, ny * 500.0
    f = idz.supercell_fields(nens, nx, ny, nz, zint, tracers=tr, magnitude=0.5)
    idz.add_tracer_blobs(f, tr, xlen, ylen, zint)
    res = []
    for graph in ("off", "on"):
        coupler = PamCoupler("cuda:0")
        coupler.set_option("crm_dt", 2.0)
        coupler.allocate_coupler_state(nz, ny, nx, nens)
        coupler.set_grid(xlen, ylen, zint)
        for n, p, m in tr:
            coupler.add_tracer(n, "", p, m)
        d = Dycore()
        d.init(coupler)
        d.set_graph_replay(graph)
        coupler.load_fields(f)
        d.declare_current_profile_as_hydrostatic(coupler)
        if graph == "on":
            d.debug_fail_next_capture(which)
        ncyc = [d.timeStep(coupler)]
        if graph == "on":
            msg = capi.load().pam_amd_awfl_last_error().decode()
            assert msg.startswith("warning: time_step:") and "eagerly" in msg, msg
        ncyc += [d.timeStep(coupler) for _ in range(2)]      # the handle keeps working (eagerly) afterwards
        torch.cuda.synchronize()
        res.append((ncyc, coupler.dump_fields()))
        d.finalize(coupler)
    assert res[0][0] == res[1][0]
    for k in res[0][1]:
        assert np.isfinite(res[0][1][k]).all()
        assert np.array_equal(res[0][1][k], res[1][1][k]), k


def test_set_x_tile_is_accepted_whatever_kernels_the_handle_runs():
    """the setter applies the LDS bound of the launch (x tile kernels exchanging through LDS) to the RESOLVED mapping and
    rolls back on refusal; with sweep kernels, or a geometry the launch accepts, it must simply take the geometry"""
    from pam_amd import Dycore, PamCoupler
    coupler = PamCoupler("cuda:0")
    coupler.set_option("crm_dt", 2.0)
    coupler.allocate_coupler_state(6, 3, 64, 16)              # 64-cell lines of 16 members
    coupler.set_grid(32000.0, 1500.0, idz.uniform_interfaces(6, 6000.0))
    coupler.add_tracer("water_vapor", "", True, True)
    d = Dycore()
    d.init(coupler)
    d.set_lane_mapping("member", "sweep")
    d.set_x_tile(16, 62, 0)                                    # sweep kernels: the tile geometry is not in use
    assert not d.get_lane_mapping()["x_tiles"]
    d.set_lane_mapping("flat", "tile")
    d.set_x_tile(16, 62, 0)                                    # (62 + 2) rows x 16 lanes: 1024 lanes, 7 x 2 x 1024 doubles of LDS
    m = d.get_lane_mapping()
    assert m["x_tiles"] and m["tile"]["W"] == 16 and m["tile"]["tc"] == 62 and m["tile"]["halo"] == 1, m
    d.finalize(coupler)
