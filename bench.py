#!/usr/bin/env python3
"""bench.py -- cell-updates/s of the MI355X-native AWFL dycore step (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--nens E] [--config c2|c3|c4]

A "step" is one `Dycore::timeStep` (Dycore.h:107) over the whole resident ensemble: coupler->dycore conversion,
CFL reduction, `ncycles` SSPRK3 sub-steps (3 tendency evaluations each) and dycore->coupler conversion, with the
coupler fields already resident in HBM.  A "cell-update" is one grid cell advanced by one sub-step;
value = nens*nz*ny*nx*sum(ncycles) / wall seconds of the K timed steps (max over ranks).

Workload (config c2 = BASELINE.json configs[1], the configuration the metric is quoted on): AWFL supercell,
nens=1024 CRMs of 32x32x60 on the L60 stretched grid, NT=1 (water_vapor), crm_dt=2 s (ncycles ~ 9), synthetic
supercell sounding + splitmix64 temperature perturbation (no datasets exist offline).  At N>1 each rank owns
nens=1024 members (weak scaling by nens sharding); the only inter-rank exchange is the 8-byte
all-reduce(MIN) of the CFL time step per timeStep.

Extra objects on the JSON line:
  roofline     dominant kernel = awfl_flux_kernel (reconstruction + fluxes, ~90% of device time).  achieved =
               algorithmic bytes per launch / mean launch duration; one launch = one tendency stage over all cells =
               cells/3 cell-updates x 64*(5+NT) B.  Launch durations are measured live with HIP events on
               the stream the kernels run on (inside libpam_amd_awfl.so: pam_amd_awfl_set_kernel_timing), in a separate
               un-timed pass.  The kernel is FP64-VALU-bound, not HBM-bound: `valu` gives that roofline.
  cpu_baseline the CPU oracle (a port of the reference algorithm, oracle/awfl_oracle.c, OpenMP over the flux loop)
               timed on this host on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X FP64 vector peak (spec; 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz)

CONFIGS = {
    # name: (nens per GPU, nx, ny, tracer set, constants, crm_dt, description)
    "c2": (1024, 32, 32, "none", "default", 2.0, "AWFL supercell idealized, nens=%d/GPU, 32x32x60 L60, NT=1, fp64"),
    "c3": (4096, 32, 1, "kessler_shoc", "default", 2.0, "AWFL moist (4 advected tracers), nens=%d/GPU, 2-D 32x1x60 L60, fp64"),
    "c4": (512, 32, 1, "p3_shoc", "p3", 2.0, "AWFL + P3/SHOC tracer set (10 tracers), nens=%d/GPU, 2-D 32x1x60 L60, fp64"),
}


def make_inputs(idz, nens, nx, ny, nz, zint, tracers, consts, xlen, ylen, nens_gen=16, id0=0):
    """numpy coupler fields for nens_gen distinct members; the caller tiles them over nens on the GPU."""
    f = idz.supercell_fields(nens_gen, nx, ny, nz, zint, consts=consts, tracers=tracers, magnitude=0.1, id0=id0)
    if len(tracers) > 1:
        idz.add_tracer_blobs(f, tracers, xlen, ylen, zint)
    return f


def cpu_baseline(idz, cfg_name, nz, zint, tracers, consts, xlen, ylen, crm_dt):
    """Oracle timed on the host cores on a bounded sample (a few ensemble members of the same grid)."""
    from oracle import awfl_oracle as ao
    nens_pg, nx, ny = CONFIGS[cfg_name][:3]
    names, pos, mass, idwv = idz.tracer_flags(tracers)
    lib, kind_note = None, "generic -O2 build"
    try:   # native-tuned build of the same source for a fair CPU number
        out = os.path.join("/tmp", "libawfl_oracle_native_%d.so" % os.getpid())
        ao.build(out=out, archflags="-O3 -march=native")
        lib = ao.load(out)
        kind_note = "gcc -O3 -march=native -ffp-contract=off"
    except Exception:
        lib = ao.load()
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:   # a container's CPU quota (cgroup v2) is the real core count when it is below the affinity mask
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    threads = int(os.environ.get("OMP_NUM_THREADS", cores))
    try:
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(threads)
    except Exception:
        pass
    nens = 64 if ny > 1 else 1024      # ~10-30 s of host work on the GPU box's cores
    f = idz.supercell_fields(nens, nx, ny, nz, zint, consts=consts, tracers=tracers, magnitude=0.1)
    if len(tracers) > 1:
        idz.add_tracer_blobs(f, tracers, xlen, ylen, zint)
    o = ao.OracleDycore(nens, nx, ny, nz, xlen, ylen, np.diff(zint), pos, mass, idwv, consts=consts, lib=lib)
    o.declare_current_profile_as_hydrostatic(f)
    dt = 2.0
    o.time_step(copy.deepcopy(f), 0.2)      # warm-up (thread pool, page faults)
    t0 = time.time()
    ncyc, _ = o.time_step(f, dt)
    el = time.time() - t0
    upd = nens * nz * ny * nx * ncyc
    return {"value": upd / el, "unit": "cell-updates/s", "cores": threads, "kind": "port",
            "sample": "oracle/awfl_oracle.c (%s, OpenMP %d threads), nens=%d of the same %dx%dx%d grid, one timeStep of "
                      "crm_dt=%.2g s = %d sub-steps, %.1f s wall" % (kind_note, threads, nens, nx, ny, nz, dt, ncyc, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--nens", type=int, default=0, help="members per GPU (default: the config's)")
    ap.add_argument("--seg", type=int, default=0, help="flux-kernel chunk length (default: library default)")
    ap.add_argument("--span", type=int, default=-1, help="flux-kernel faces per thread (default: automatic)")
    ap.add_argument("--chunks", type=int, default=-1, help="internal ensemble chunks / HIP streams (default: automatic)")
    ap.add_argument("--lds-floor", type=int, default=64 * 1024, help="flux-kernel LDS floor in bytes when chunks > 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from pam_amd import Dycore, PamCoupler, idealized as idz, parallel

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # PAM_AMD_DIST_BACKEND=gloo is a rehearsal switch: several ranks on ONE GPU (RCCL refuses that), collectives on CPU
    backend = os.environ.get("PAM_AMD_DIST_BACKEND", "nccl")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dev_index = local_rank % torch.cuda.device_count()
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    nens_pg, nx, ny, trname, cname, crm_dt, desc = CONFIGS[args.config]
    if args.nens > 0:
        nens_pg = args.nens
    nz = 60
    tracers = {"none": idz.TRACERS_NONE, "kessler_shoc": idz.TRACERS_KESSLER_SHOC, "p3_shoc": idz.TRACERS_P3_SHOC}[trname]
    consts = {"default": idz.CONSTS_DEFAULT, "p3": idz.CONSTS_P3}[cname]
    nt = len(tracers)
    zint = idz.l60_interfaces()
    xlen = nx * 1000.0
    ylen = ny * 1000.0 if ny > 1 else xlen

    # ---- coupler + dycore, inputs resident in HBM before anything is timed
    coupler = PamCoupler(dev)
    coupler.set_option("crm_dt", crm_dt)
    for k, v in consts.items():
        coupler.set_option(k, v)
    coupler.allocate_coupler_state(nz, ny, nx, nens_pg)
    coupler.set_grid(xlen, ylen, zint)
    for n, p, m in tracers:
        coupler.add_tracer(n, "", p, m)
    dycore = Dycore()
    dycore.init(coupler)
    if args.seg > 0:
        dycore.set_flux_segment(args.seg)
    if args.span >= 0:
        dycore.set_flux_span(args.span)
    if args.chunks >= 0:
        dycore.set_ensemble_chunks(args.chunks, args.lds_floor)
    nens_gen = min(16, nens_pg)
    f = make_inputs(idz, nens_pg, nx, ny, nz, zint, tracers, consts, xlen, ylen, nens_gen=nens_gen, id0=rank * 1000)
    reps = (nens_pg + nens_gen - 1) // nens_gen
    # members differ between tiles by a small smooth temperature offset so no two CRMs are identical
    off = (torch.arange(nens_pg, device=dev, dtype=torch.float64) // nens_gen) * 1.0e-3

    def tile(a):
        return torch.from_numpy(a).to(dev).repeat(*([1] * (a.ndim - 1)), reps)[..., :nens_pg].contiguous()
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp"):
        coupler.dm.get(k).copy_(tile(f[k]))
    coupler.dm.get("temp").add_(off)
    for t, name in enumerate(coupler.get_tracer_names()):
        coupler.dm.get(name).copy_(tile(f["tracers"][t]))
    del f
    dycore.declare_current_profile_as_hydrostatic(coupler)
    torch.cuda.synchronize()

    def one_step():
        if world > 1:
            return parallel.sharded_time_step(dycore, coupler)
        return dycore.timeStep(coupler)

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    substeps = 0
    for _ in range(args.steps):
        substeps += one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    cells = nens_pg * nz * ny * nx
    total_updates = cells * substeps * world
    value = total_updates / elapsed

    # ---- per-kernel durations (HIP events on the stream each kernel is launched on), separate un-timed passes
    roofline = None
    kernels = {}
    if not args.no_kernel_timing:
        def timed_pass():
            dycore.reset_kernel_timing()
            one_step()
            torch.cuda.synchronize()
            out = {}
            for name in ("flux", "fct_mult", "update", "init_prim", "finalize", "cfl"):
                ms, n = dycore.get_kernel_timing(name)
                if n:
                    out[name] = {"launches": n, "avg_ms": ms / n, "total_ms": ms}
            return out
        dycore.set_kernel_timing(True)
        kernels = timed_pass()                         # shipped configuration (chunks overlap: durations include contention)
        # kernel-level roofline: the dominant kernel on its own (one chunk = whole ensemble per launch, nothing co-running)
        dycore.set_ensemble_chunks(1)
        alone = timed_pass()
        dycore.set_ensemble_chunks(args.chunks if args.chunks >= 0 else 0, args.lds_floor)
        dycore.set_kernel_timing(False)
        if "flux" in alone:
            avg_s = alone["flux"]["avg_ms"] * 1e-3
            alg_bytes = cells / 3.0 * 64.0 * (5 + nt)            # 64*(5+NT) B per cell-update, 1/3 per stage
            achieved = alg_bytes / avg_s / 1e9
            ndir = 2 if ny == 1 else 3
            # polynomials per cell-stage: ndir x (2 acoustic + 4+NT advected), x(span+1)/span (whole-line spans: ~1.03);
            # 119 (uniform-grid directions) / 128 (vertical) FP64 instructions per polynomial of which half are FMAs (ISA
            # count of the inner loops) -> ~183 flop on average
            flops = cells * ndir * (6 + nt) * 1.03 * 183.0
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "r01_c2_flux_traffic.json")
            if args.config == "c2" and args.nens == 0 and os.path.exists(tpath):
                traffic = json.load(open(tpath))["hbm_bytes_per_launch"]   # rocprofv3 PMC passes of this same command
            roofline = {"bound": "hbm", "kernel": "awfl_flux_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                        "avg_launch_ms": alone["flux"]["avg_ms"], "alg_bytes_per_launch": alg_bytes,
                        "note": "one launch = one tendency stage of the whole ensemble, measured with --chunks 1 "
                                "(nothing co-running); in the shipped chunked configuration launches are per chunk and "
                                "overlap the update kernels (see kernels)",
                        "valu": {"bound": "fp64-valu", "achieved": flops / avg_s / 1e12, "peak": FP64_VALU_PEAK_TFLOPS,
                                 "unit": "TFLOP/s", "frac": flops / avg_s / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                 "note": "the kernel is FP64-VALU-bound (SURVEY F5; counters: SQ_INSTS_VALU x 4 cycles "
                                         "over GRBM_GUI_ACTIVE -> VALU issuing in ~75-79% of the cycles at the ~2.1 GHz clock "
                                         "held; profiles/r01_c2_pmc_summary.txt): this is the roofline that binds"}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(idz, args.config, nz, zint, tracers, consts, xlen, ylen, crm_dt)
        except Exception as e:   # the baseline is a reported extra; never fail the bench line for it
            cpu = {"value": None, "unit": "cell-updates/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}

    if rank == 0:
        out = {"metric": "cell-updates/sec (AWFL dycore step)", "value": value, "unit": "cell-updates/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": desc % nens_pg, "nens_per_gpu": nens_pg, "nens_total": nens_pg * world, "nx": nx, "ny": ny,
                          "nz": nz, "num_tracers": nt, "crm_dt": crm_dt, "substeps_per_step": substeps / args.steps,
                          "parallelism": "nens-shard x%d" % world},
               "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels}
        print(json.dumps(out))
    dycore.finalize(coupler)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
