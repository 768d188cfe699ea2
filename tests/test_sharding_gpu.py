"""GPU: the nens-sharded path through the HIP library itself.  One ensemble is split into two couplers on
the same GPU (what two ranks of an 8-GPU node hold); both are stepped with dt_dyn_hint = min(dt_0, dt_1) -- the 8-byte
all-reduce(MIN) of pam_amd.parallel.sharded_time_step -- and the concatenation must equal the unsharded HIP run bit for bit
(reference semantics: dt_dyn is a minimum over ALL members, Dycore.h:86-101,141-145).  Without the exchange the shards
sub-cycle differently when one of them holds the CFL-limiting member.  (tests/test_sharding_gloo.py proves the same
contract with real processes over gloo, stepping the oracle.)"""
import numpy as np
import pytest

from pam_amd import idealized as idz

pytestmark = pytest.mark.gpu


def _mk(f, lo, hi, nx, ny, nz, zint, tr):
    from pam_amd import Dycore, PamCoupler
    nens = hi - lo
    c = PamCoupler("cuda:0")
    c.set_option("crm_dt", 1.0)
    c.allocate_coupler_state(nz, ny, nx, nens)
    c.set_grid(nx * 500.0, (ny if ny > 1 else nx) * 500.0, zint)
    for n, p, m in tr:
        c.add_tracer(n, "", p, m)
    d = Dycore()
    d.init(c)
    c.load_fields({k: np.ascontiguousarray(v[..., lo:hi]) for k, v in f.items()})
    d.declare_current_profile_as_hydrostatic(c)
    return c, d


def test_two_shards_with_reduced_dt_equal_the_unsharded_run_bit_for_bit():
    import torch
    from pam_amd import parallel
    nens, nx, ny, nz = 6, 8, 4, 10
    tr = idz.TRACERS_KESSLER_SHOC
    zint = idz.stretched_interfaces(nz, 12000.0)
    f = idz.supercell_fields(nens, nx, ny, nz, zint, tracers=tr, magnitude=0.5)
    idz.add_tracer_blobs(f, tr, nx * 500.0, ny * 500.0, zint)
    f["uvel"][..., 5] += 60.0                      # member 5 (second shard) limits the CFL step of the whole ensemble
    whole_c, whole_d = _mk(f, 0, nens, nx, ny, nz, zint, tr)
    ranges = [parallel.shard_range(nens, r, 2) for r in range(2)]
    assert ranges == [(0, 3), (3, 6)]
    shards = [_mk(f, lo, hi, nx, ny, nz, zint, tr) for lo, hi in ranges]
    free = [_mk(f, lo, hi, nx, ny, nz, zint, tr) for lo, hi in ranges]
    for _ in range(2):
        n_whole = whole_d.timeStep(whole_c)
        dts = [d.compute_time_step(c) for c, d in shards]
        assert dts[1] < dts[0]
        dt = min(dts)                               # == parallel.global_min over the ranks
        n_sh = [d.timeStep(c, dt_dyn_hint=dt) for c, d in shards]
        n_free = [d.timeStep(c) for c, d in free]   # every shard on its own CFL step: NOT the reference semantics
        assert n_sh == [n_whole, n_whole]
    assert n_free[0] < n_whole and n_free[1] == n_whole
    torch.cuda.synchronize()
    w = whole_c.dump_fields()
    s = [c.dump_fields() for c, d in shards]
    fr = [c.dump_fields() for c, d in free]
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
        cat = np.concatenate([s[0][k], s[1][k]], axis=-1)
        assert np.array_equal(cat, w[k]), k
    assert not np.array_equal(fr[0]["temp"], w["temp"][..., :3])      # without the exchange the first shard differs
    assert np.array_equal(fr[1]["temp"], w["temp"][..., 3:])          # (the limiting shard happens to agree)
    for c, d in [(whole_c, whole_d)] + shards + free:
        d.finalize(c)


def test_sharded_time_step_helper_without_process_group_is_the_local_step():
    """pam_amd.parallel.sharded_time_step on a single process (no torch.distributed group): the global minimum is the local
    one and the result equals Dycore.timeStep."""
    import torch
    from pam_amd import parallel
    nens, nx, ny, nz = 3, 8, 1, 10
    tr = idz.TRACERS_NONE
    zint = idz.uniform_interfaces(nz, 10000.0)
    f = idz.supercell_fields(nens, nx, ny, nz, zint, tracers=tr, magnitude=0.5)
    a_c, a_d = _mk(f, 0, nens, nx, ny, nz, zint, tr)
    b_c, b_d = _mk(f, 0, nens, nx, ny, nz, zint, tr)
    na = a_d.timeStep(a_c)
    nb = parallel.sharded_time_step(b_d, b_c)
    torch.cuda.synchronize()
    assert na == nb
    x, y = a_c.dump_fields(), b_c.dump_fields()
    for k in x:
        assert np.array_equal(x[k], y[k]), k
    a_d.finalize(a_c)
    b_d.finalize(b_c)


def _mk_large(nens, nt_set, seed_id=0):
    """an ensemble large enough for member lanes and two independent member ranges (>= 128 members)"""
    nx, ny, nz = 16, 1, 12
    tr = nt_set
    zint = idz.stretched_interfaces(nz, 12000.0)
    f = idz.supercell_fields(16, nx, ny, nz, zint, tracers=tr, magnitude=0.5, id0=seed_id)
    if len(tr) > 1:
        idz.add_tracer_blobs(f, tr, nx * 500.0, nx * 500.0, zint)
    f = {k: np.ascontiguousarray(np.tile(v, (1,) * (v.ndim - 1) + (nens // 16,))) for k, v in f.items()}
    f["temp"] = f["temp"] + 1e-3 * (np.arange(nens) // 16)
    return _mk(f, 0, nens, nx, ny, nz, zint, tr)


def test_two_handles_in_one_process_tuned_differently_do_not_reshape_each_other():
    """the launch-shape thresholds are per handle.  Two handles of one process (what examples/driver --gpus N holds:
    one per device) get DIFFERENT thresholds -- and the process-wide default is changed after both exist -- and both must still
    produce the bits of an untuned handle (results never depend on launch shapes; before round 5 the thresholds were three
    process-global statics, so tuning one handle silently re-shaped the launches of every other, unsynchronised)."""
    import torch
    from pam_amd import capi
    tr = idz.TRACERS_KESSLER_SHOC
    ref_c, ref_d = _mk_large(128, tr)
    a_c, a_d = _mk_large(128, tr)
    b_c, b_d = _mk_large(128, tr)
    a_d.set_launch_tuning(64, 0, 0)                  # whole lines, one-phase y/z sweeps, tracer phase 1 inline
    b_d.set_launch_tuning(1 << 20, 1 << 30, 1 << 30)   # shortest spans, two-phase sweeps, phase 1 in its own launch
    lib = capi.load()
    capi.check(lib.pam_amd_awfl_set_launch_tuning(777, 5, 5))     # the defaults of handles created from now on: a, b, ref keep theirs
    try:
        for _ in range(2):
            n = [d.timeStep(c) for c, d in ((ref_c, ref_d), (a_c, a_d), (b_c, b_d))]
            assert n[0] == n[1] == n[2]
        torch.cuda.synchronize()
        r, a, b = ref_c.dump_fields(), a_c.dump_fields(), b_c.dump_fields()
        for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
            assert np.array_equal(a[k], r[k]) and np.array_equal(b[k], r[k]), k
        # the two tunings really are different launch shapes: kernel launch counts per step differ
        for d in (a_d, b_d):
            d.set_kernel_timing(True)
            d.reset_kernel_timing()
        a_d.timeStep(a_c)
        b_d.timeStep(b_c)
        torch.cuda.synchronize()
        assert a_d.get_kernel_timing("xtr1")[1] == 0 and b_d.get_kernel_timing("xtr1")[1] > 0
    finally:
        capi.check(lib.pam_amd_awfl_set_launch_tuning(3072, 8192, 8192))
    for c, d in ((ref_c, ref_d), (a_c, a_d), (b_c, b_d)):
        d.finalize(c)


def test_a_handle_created_beside_eight_idle_streams_gives_the_two_range_result_bit_for_bit():
    """streams map onto a handful of hardware queues, and a host model with streams of its own may push the
    handle's two member ranges onto one queue (slower, never different): a handle created while 8 unrelated idle streams exist, and
    one restricted to a single range (the robust setting of INTEGRATION.md section 4), equal the plain two-range run bit for bit."""
    import torch
    tr = idz.TRACERS_NONE
    ref_c, ref_d = _mk_large(256, tr)
    streams = [torch.cuda.Stream() for _ in range(8)]
    busy_c, busy_d = _mk_large(256, tr)
    one_c, one_d = _mk_large(256, tr)
    one_d.set_ensemble_chunks(1, 0)
    for _ in range(2):
        n = [d.timeStep(c) for c, d in ((ref_c, ref_d), (busy_c, busy_d), (one_c, one_d))]
        assert n[0] == n[1] == n[2]
    torch.cuda.synchronize()
    r, b, o = ref_c.dump_fields(), busy_c.dump_fields(), one_c.dump_fields()
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
        assert np.array_equal(b[k], r[k]) and np.array_equal(o[k], r[k]), k
    del streams
    for c, d in ((ref_c, ref_d), (busy_c, busy_d), (one_c, one_d)):
        d.finalize(c)


@pytest.mark.parametrize("trname", ["p3_shoc", "kessler_shoc", "three", "six"])
def test_four_tracers_per_wavefront_equal_pairs_bit_for_bit(trname):
    """pam_amd_awfl_set_tracer_grouping (VERDICT r3 / r4 experiment (a)): the separately launched x sweeps of the further tracers with up to
    FOUR tracers per wavefront (groups of 4 + a remainder of 1, 2 or 3) against pairs: same arithmetic per tracer, same bits.  9, 3, 2 and
    5 further tracers: remainders 1, 3, 2 and 1."""
    import torch
    tr = {"p3_shoc": idz.TRACERS_P3_SHOC, "kessler_shoc": idz.TRACERS_KESSLER_SHOC,
          "three": (("cloud", True, True), ("water_vapor", True, True), ("rain", True, False)),
          "six": tuple([("t%d" % i, True, i % 2 == 0) for i in range(5)] + [("water_vapor", True, True)])}[trname]
    a_c, a_d = _mk_large(128, tr)
    b_c, b_d = _mk_large(128, tr)
    p_c, p_d = _mk_large(128, tr)
    s_c, s_d = _mk_large(128, tr)
    for d in (a_d, b_d, p_d, s_d):
        d.set_launch_tuning(0, -1, 1 << 30)          # phase 1 of the tracer sweeps as a launch of its own
    a_d.set_tracer_grouping(2)                       # pairs in both phases (the round-3 / round-4 form)
    b_d.set_tracer_grouping(4)
    p_d.set_tracer_grouping(2, prefetch=True)        # experiment (b): phase 2 with the next trip's loads one trip ahead
    s_d.set_tracer_grouping(0)                       # automatic: phase 1 one tracer per wavefront, phase 2 pairs (three further tracers: singles)
    for _ in range(2):
        assert a_d.timeStep(a_c) == b_d.timeStep(b_c) == p_d.timeStep(p_c) == s_d.timeStep(s_c)
    torch.cuda.synchronize()
    a, b, pf, sg = a_c.dump_fields(), b_c.dump_fields(), p_c.dump_fields(), s_c.dump_fields()
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
        assert np.isfinite(a[k]).all() and np.array_equal(a[k], b[k]) and np.array_equal(a[k], pf[k]) and np.array_equal(a[k], sg[k]), k
    for c, d in ((a_c, a_d), (b_c, b_d), (p_c, p_d), (s_c, s_d)):
        d.finalize(c)


FUZZ_SEEDS = int(__import__("os").environ.get("PAM_AMD_FUZZ_SEEDS", "10"))


@pytest.mark.parametrize("seed", range(FUZZ_SEEDS))
def test_random_shards_equal_the_unsharded_run_bit_for_bit(seed):
    """seeded random ensembles cut at random member indices into 2 .. 5 shards of ANY size (one member, ragged, across the 64-member
    boundary where a shard switches from member lanes to tile kernels): with the dt minimum exchanged, the concatenation equals the
    unsharded run bit for bit -- the lane mapping a shard resolves for ITS member count never shows in the result
    (PAM_AMD_FUZZ_SEEDS=N: N seeds)"""
    import torch
    rng = np.random.default_rng(50021 * seed + 11)
    while True:
        nens = int(rng.choice([2, 3, 5, 8, 17, 40, 64, 65, 100, 130, 200]))
        nx, ny, nz = int(rng.integers(3, 33)), int(rng.choice([1, 1, 3, 5])), int(rng.integers(4, 17))
        if nens * nx * ny * nz <= 300000:
            break
    tr = [idz.TRACERS_NONE, idz.TRACERS_KESSLER_SHOC, idz.TRACERS_P3_SHOC][int(rng.integers(0, 3))]
    zint = idz.stretched_interfaces(nz, 11000.0)
    f = idz.supercell_fields(nens, nx, ny, nz, zint, tracers=tr, magnitude=0.5, id0=seed)
    idz.add_tracer_blobs(f, tr, nx * 500.0, (ny if ny > 1 else nx) * 500.0, zint)
    if rng.random() < 0.5:
        idz.carve_dry_air(f, tr)
    f["uvel"][..., int(rng.integers(0, nens))] += 40.0          # some member limits the CFL step of the whole ensemble
    nshards = int(rng.integers(2, min(5, nens) + 1))
    cuts = sorted(int(c) for c in rng.choice(np.arange(1, nens), size=nshards - 1, replace=False))
    ranges = list(zip([0] + cuts, cuts + [nens]))
    what = "seed %d: nens %d, %dx%dx%d, nt %d, shards %s" % (seed, nens, nx, ny, nz, len(tr), ranges)
    whole_c, whole_d = _mk(f, 0, nens, nx, ny, nz, zint, tr)
    shards = [_mk(f, lo, hi, nx, ny, nz, zint, tr) for lo, hi in ranges]
    for _ in range(2):
        n_whole = whole_d.timeStep(whole_c)
        dt = min(d.compute_time_step(c) for c, d in shards)
        assert [d.timeStep(c, dt_dyn_hint=dt) for c, d in shards] == [n_whole] * nshards, what
    torch.cuda.synchronize()
    w = whole_c.dump_fields()
    s = [c.dump_fields() for c, d in shards]
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
        assert np.isfinite(w[k]).all(), (what, k)
        assert np.array_equal(np.concatenate([x[k] for x in s], axis=-1), w[k]), (what, k)
    for c, d in [(whole_c, whole_d)] + shards:
        d.finalize(c)
