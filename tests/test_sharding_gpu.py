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
