"""N>1 path on CPU: nens sharding + the one exchange step (all-reduce MIN of the CFL time step) over gloo,
world_size 2.  Each rank advances its shard with the oracle (the GPU kernels cannot run here); the property checked
is the sharding contract of pam_amd.parallel: with the globally reduced dt every shard reproduces the unsharded
run bit for bit, and without it the shards sub-cycle differently."""
import copy
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case():
    from pam_amd import idealized as idz
    nens, nx, ny, nz = 6, 8, 1, 10
    zint = idz.uniform_interfaces(nz, 10000.0)
    f = idz.dry_bubble_fields(nens, nx, ny, nz, 8000.0, 8000.0, zint, amp0=2.0, damp=0.5)
    # make member 5 the CFL-limiting one, so the two shards disagree on dt without the reduction
    f["uvel"][..., 5] += 60.0
    return nens, nx, ny, nz, zint, f


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import awfl_oracle as ao
    from pam_amd import idealized as idz, parallel
    nens, nx, ny, nz, zint, f = _case()
    lo, hi = parallel.shard_range(nens, rank, world)
    fs = {k: np.ascontiguousarray(v[..., lo:hi]) for k, v in f.items()}
    names, pos, mass, idwv = idz.tracer_flags(idz.TRACERS_NONE)
    o = ao.OracleDycore(hi - lo, nx, ny, nz, 8000.0, 8000.0, np.diff(zint), pos, mass, idwv)
    o.declare_current_profile_as_hydrostatic(fs)
    dt_local = o.compute_time_step(fs)
    dt = parallel.global_min(dt_local)
    ncyc, _ = o.time_step(fs, 1.0, dt_dyn=dt)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo, hi=hi, dt_local=dt_local, dt=dt, ncyc=ncyc,
             **{k: v for k, v in fs.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_step_equals_unsharded(tmp_path):
    from oracle import awfl_oracle as ao
    from pam_amd import idealized as idz
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    nens, nx, ny, nz, zint, f = _case()
    ref = copy.deepcopy(f)
    names, pos, mass, idwv = idz.tracer_flags(idz.TRACERS_NONE)
    o = ao.OracleDycore(nens, nx, ny, nz, 8000.0, 8000.0, np.diff(zint), pos, mass, idwv)
    o.declare_current_profile_as_hydrostatic(ref)
    dt_ref = o.compute_time_step(ref)
    ncyc_ref, _ = o.time_step(ref, 1.0)
    parts = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    assert [int(p["lo"]) for p in parts] == [0, 3] and [int(p["hi"]) for p in parts] == [3, 6]
    assert float(parts[0]["dt_local"]) > float(parts[1]["dt_local"])            # shards disagree locally ...
    assert all(float(p["dt"]) == dt_ref for p in parts)                         # ... and agree after the MIN
    assert all(int(p["ncyc"]) == ncyc_ref for p in parts)
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp", "tracers"):
        got = np.concatenate([p[k] for p in parts], axis=-1)
        assert np.array_equal(got, ref[k]), k


def test_shard_ranges_partition_the_ensemble():
    from pam_amd import parallel
    for n in (1, 7, 8, 1024, 1030):
        for w in (1, 2, 4, 8):
            r = [parallel.shard_range(n, i, w) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
