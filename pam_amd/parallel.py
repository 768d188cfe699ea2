"""nens sharding of the CRM ensemble across the GPUs of one node.

Every AWFL kernel is independent across ensemble members (periodic halos are intra-CRM, Dycore.h:629-657; dz,
vertical matrices and gravity are per member), so rank r simply owns a contiguous block of members and holds its
own (nz,ny,nx,nens_local) arrays.  The ONE exchange the reference semantics need is the sub-cycling time step:
`dt_dyn` is a minimum over ALL members (Dycore.h:86-101) and `ncycles = ceil(crm_dt/dt_dyn)` (Dycore.h:144-145),
so shards must agree on min_r(dt_r): one 8-byte all-reduce(MIN) per timeStep (RCCL over xGMI on GPUs, gloo in the
CPU tests) -- latency-only, off the per-stage path.  No halo, no other collective.
"""
import torch
import torch.distributed as dist


def shard_range(nens_total, rank, world_size):
    """Members [lo, hi) owned by `rank`; blocks differ by at most one member."""
    base, rem = divmod(nens_total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_MIN_BUF = {}


def global_min(value, device=None, group=None):
    """min over ranks of a python float (identity when torch.distributed is not initialised).  ONE reusable 8-byte tensor per device
    (pinned when it lives on the host): no allocation per timeStep."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(value)
    key = str(device) if device is not None else "cpu"
    t = _MIN_BUF.get(key)
    if t is None:
        if device is not None:
            t = torch.empty(1, dtype=torch.float64, device=device)
        else:
            t = torch.empty(1, dtype=torch.float64)
            try:
                t = t.pin_memory()
            except Exception:
                pass
        _MIN_BUF[key] = t
    t.fill_(float(value))
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return float(t.item())


def sharded_time_step(dycore, coupler, cfl=0.8, group=None):
    """Dycore::timeStep on this rank's shard with the ensemble-global dt_dyn (what the unsharded reference computes): the shard's
    CFL minimum (one 8-byte read-back: compute_time_step), ONE all-reduce(MIN) of 8 bytes, then time_step with the agreed dt as a
    hint -- no second read-back, no further host synchronisation until the caller asks for one."""
    dt_local = dycore.compute_time_step(coupler, cfl)
    backend = dist.get_backend(group) if (dist.is_available() and dist.is_initialized()) else None
    dev = coupler.device if backend == "nccl" else None
    dt = global_min(dt_local, device=dev, group=group)
    return dycore.timeStep(coupler, dt_dyn_hint=dt)
