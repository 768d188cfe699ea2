#!/usr/bin/env python3
"""bench.py -- cell-updates/s of the MI355X-native AWFL dycore step (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4] [--scaling weak|strong] [--nens E]

A "step" is one `Dycore::timeStep` (Dycore.h:107) over the whole resident ensemble: coupler->dycore conversion,
CFL reduction, `ncycles` SSPRK3 sub-steps (3 tendency evaluations each) and dycore->coupler conversion, with the
coupler fields already resident in HBM.  A "cell-update" is one grid cell advanced by one sub-step;
value = sum over ranks of nens*nz*ny*nx*sum(ncycles) / wall seconds of the K timed steps (max over ranks).

Workload (config c2 = BASELINE.json configs[1], the configuration the metric is quoted on): AWFL supercell,
nens=1024 CRMs of 32x32x60 on the L60 stretched grid, NT=1 (water_vapor), crm_dt=2 s (ncycles ~ 9), synthetic
supercell sounding + splitmix64 temperature perturbation (no datasets exist offline).

Ranks.  One process per GPU.  Under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks
come from RANK/LOCAL_RANK/WORLD_SIZE (WORLD_SIZE must equal --gpus).  Without those variables `--gpus N` makes THIS
process a launcher: it starts N fresh child processes (before anything touches a GPU: the parent never imports torch),
one rank each, and prints rank 0's JSON line.  The ensemble shards by member index with no data-path collective; the one
exchange is the 8-byte all-reduce(MIN) of the CFL time step per timeStep (RCCL; Dycore.h:141-145 semantics).
`--scaling weak` (default): every rank holds the config's nens (1024 for c2).  `--scaling strong`: the config's nens is
the TOTAL, split over the ranks (c2: 1024 -> 128 per GPU at N=8; c4's 512 is already the per-GPU shard of nens=4096).
When the box has fewer GPUs than ranks (rehearsal on a 1-GPU box) the ranks share device 0 and reduce over gloo.

Extra objects on the JSON line:
  roofline      the dominant kernel of the stage (by device time, nothing co-running): achieved = algorithmic bytes per
                launch / mean launch duration; one launch = one tendency stage over all cells = cells/3 cell-updates x
                64*(5+NT) B.  Launch durations are measured live with HIP events on the stream the kernels
                run on (inside libpam_amd_awfl.so: pam_amd_awfl_set_kernel_timing), in a separate un-timed pass.
                achieved / peak / frac (= hbm_frac) are the HBM view the metric asks for; `bound` names the roofline that
                actually limits the kernel ("fp64-valu" when its VALU fraction exceeds its HBM fraction) and `valu`
                quantifies it from the profile's SQ_INSTS_VALU / GRBM_GUI_ACTIVE (issue fraction at the clock the chip
                held, and fraction of the 2.4 GHz spec with this run's duration).
                `traffic` = HBM bytes per stage from rocprofv3 PMC passes of this command, valid only for the build they
                were taken from (content hash of pam_amd/csrc; null when the sources have changed since);
                `stage_traffic_ratio` = HBM bytes moved by all stage kernels / algorithmic bytes of the stage.
                `kernel_rooflines` lists every stage kernel with the bytes it must itself move and its FP64 work.
  cpu_baseline  the CPU oracle (a port of the reference algorithm, oracle/awfl_oracle.c, OpenMP over the flux loop)
                timed on this host on a bounded sample of the same workload (rank 0, N=1 only).
  other_configs C3 / C4 throughput on this GPU (N=1, default run only), measured the same way with fewer steps, each with its
                own `roofline` (+ `traffic` from profiles/r03_c{3,4}_traffic.json) and `kernel_rooflines`.
"""
import argparse
import copy
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X FP64 vector peak (spec; 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz)

CONFIGS = {
    # name: (nens per GPU, nx, ny, tracer set, constants, crm_dt, description[, nz, vertical grid, xlen])
    "c2": (1024, 32, 32, "none", "default", 2.0, "AWFL supercell idealized, nens=%d/GPU, 32x32x60 L60, NT=1, fp64"),
    "c3": (4096, 32, 1, "kessler_shoc", "default", 2.0, "AWFL moist (4 advected tracers), nens=%d/GPU, 2-D 32x1x60 L60, fp64"),
    "c4": (512, 32, 1, "p3_shoc", "p3", 2.0, "AWFL + P3/SHOC tracer set (10 tracers), nens=%d/GPU, 2-D 32x1x60 L60, fp64"),
    # C2's 3-D grid with the Kessler + SHOC tracer set (not a BASELINE config: the 3-D many-tracer case of the small-ensemble kernels)
    "c2k": (1024, 32, 32, "kessler_shoc", "default", 2.0, "AWFL moist 3-D (4 advected tracers), nens=%d/GPU, 32x32x60 L60, fp64"),
    # the shape of the reference's own input file (standalone/mmf_simplified/inputs/input_pama.yaml: crm_nx 250, crm_ny 1,
    # nens 1, 50 equal levels to 20 km, xlen 128 km) with the Kessler + SHOC tracer registrations
    "ref": (1, 250, 1, "kessler_shoc", "default", 2.0, "reference input shape (input_pama.yaml), nens=%d/GPU, 2-D 250x1x50 uniform 20 km, NT=4, fp64",
            50, "uniform20km", 128000.0),
}
STAGE_KERNELS = ("flux", "xupd", "xtr1", "xtr2", "ptail", "trfix", "fct_mult", "update")


def csrc_hash():
    """content hash of the DYCORE kernel sources (pam_amd/csrc/awfl_*): ties a committed PMC profile of the stage kernels to the build
    it was measured on (the coupler modules' kernels, modules_kernels.hip, are not in those profiles)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pam_amd", "csrc")
    for f in sorted(os.listdir(d)):
        p = os.path.join(d, f)
        if os.path.isfile(p) and f.startswith("awfl_") and f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n):
    """Parent of an N-rank run: fresh children, started before this process has imported torch or touched a GPU."""
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit("bench.py: rank exit codes %r" % (rcs,))


def launch_cpp(args):
    """--launcher cpp: the same workload through the C++ host path -- examples/driver --gpus N: one host thread, one coupler and one
    dycore handle per device, the dt minimum over N host doubles, no collective library and no Python in the timed region.  This
    process only writes the synthetic input (16 distinct members; the driver tiles them, +t mK on temp per tile, as Job does) and
    turns the driver's wall time into the bench line."""
    import struct
    import tempfile
    import numpy as np
    from pam_amd import idealized as idz
    nens_pg, nx, ny, trname, cname, crm_dt, desc = CONFIGS[args.config][:7]
    nz, grid, xlen_cfg = CONFIGS[args.config][7:] or (60, "l60", None)
    if args.nens > 0:
        nens_pg = args.nens
    nens_total = nens_pg if args.scaling == "strong" else nens_pg * args.gpus
    nens_gen = min(16, nens_total)
    if nens_total % nens_gen:
        raise SystemExit("bench.py --launcher cpp: the ensemble (%d) must be a multiple of %d generated members" % (nens_total, nens_gen))
    tracers = {"none": idz.TRACERS_NONE, "kessler_shoc": idz.TRACERS_KESSLER_SHOC, "p3_shoc": idz.TRACERS_P3_SHOC}[trname]
    consts = {"default": idz.CONSTS_DEFAULT, "p3": idz.CONSTS_P3}[cname]
    zint = idz.l60_interfaces() if grid == "l60" else idz.uniform_interfaces(nz, 20000.0)
    xlen = xlen_cfg if xlen_cfg else nx * 1000.0
    ylen = ny * 1000.0 if ny > 1 else xlen
    f = idz.supercell_fields(nens_gen, nx, ny, nz, zint, consts=consts, tracers=tracers, magnitude=0.1, id0=0)
    if len(tracers) > 1:
        idz.add_tracer_blobs(f, tracers, xlen, ylen, zint)
    if args.limiter:
        idz.carve_dry_air(f, tracers, spread=args.limiter > 1)
    names, pos, mass, idwv = idz.tracer_flags(tracers)
    driver = os.path.join(ROOT, "examples", "driver")
    if not os.path.exists(driver):
        raise SystemExit("bench.py --launcher cpp: examples/driver is missing (python -c 'import __graft_entry__ as g; g.build()')")
    with tempfile.TemporaryDirectory() as td:
        inp = os.path.join(td, "in.bin")
        with open(inp, "wb") as fh:
            fh.write(struct.pack("<8q", nens_gen, nx, ny, nz, len(tracers), 0, 1, 1))
            fh.write(struct.pack("<3d", xlen, ylen, crm_dt))
            fh.write(struct.pack("<6d", *[consts[k] for k in ("R_d", "cp_d", "R_v", "cp_v", "p0", "grav")]))
            fh.write(np.asarray(zint, dtype="<f8").tobytes())
            fh.write(bytes(bytearray(v for t in range(len(tracers)) for v in (int(pos[t]), int(mass[t])))))
            fh.write(struct.pack("<q", idwv))
            for k in ("density_dry", "uvel", "vvel", "wvel", "temp"):
                fh.write(f[k].astype("<f8").tobytes())
            for t in range(len(tracers)):
                fh.write(f["tracers"][t].astype("<f8").tobytes())
        r = subprocess.run([driver, "--gpus", str(args.gpus), "--tile", str(nens_total // nens_gen), "--bench", str(args.steps),
                            str(args.warmup), inp, "-"], capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit("bench.py --launcher cpp: examples/driver failed: %s" % r.stderr[-2000:])
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    updates = float(nens_total) * nz * ny * nx * d["substeps"]
    out = {"metric": "cell-updates/sec (AWFL dycore step)", "value": updates / d["seconds"], "unit": "cell-updates/s",
           "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": d["seconds"] / args.steps * 1e3,
           "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"scaling": args.scaling, "nens_total": nens_total, "nens_per_gpu": nens_total // args.gpus,
                      "workload": desc % (nens_total // args.gpus),
                      "nx": nx, "ny": ny, "nz": nz, "num_tracers": len(tracers), "crm_dt": crm_dt,
                      "substeps_per_step": d["substeps"] / float(args.steps), "parallelism": "nens-shard x%d" % args.gpus,
                      "launcher": "cpp: examples/driver --gpus %d (one host thread + one dycore handle per device; dt = min over %d host "
                                  "doubles; no collective library)" % (args.gpus, args.gpus),
                      "collective": None, "ranks_seen": d["ranks"], "devices_seen": d["devices"], "limiter_input": args.limiter},
           "roofline": None, "cpu_baseline": None,
           "note": "roofline / cpu_baseline are measured by the default (Python) launcher of the same library"}
    emit(out, args)


def _host_cores():
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
    return cores


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown CPU"


def _oracle_lib():
    """the oracle built with -O3 -march=native for a fair CPU number (once per run; shared by the worker processes)"""
    from oracle import awfl_oracle as ao
    out = os.path.join("/tmp", "libawfl_oracle_native_%d.so" % os.getuid())
    try:
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(os.path.join(ROOT, "oracle", "awfl_oracle.c")):
            ao.build(out=out, archflags="-O3 -march=native")
        return ao.load(out), "gcc -O3 -march=native -ffp-contract=off"
    except Exception:
        return ao.load(), "generic -O2 build"


def cpu_worker(cfg_name, nens, threads):
    """one timed oracle timeStep on `nens` members of the config's grid with `threads` OpenMP threads -> cell-updates/s
    (a process of its own: `python bench.py --cpu-worker cfg,nens,threads`)"""
    import numpy as np
    from oracle import awfl_oracle as ao
    import importlib.util     # pam_amd/idealized.py is numpy-only: load it by path, without the package (which imports torch)
    spec = importlib.util.spec_from_file_location("pam_amd_idealized", os.path.join(ROOT, "pam_amd", "idealized.py"))
    idz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(idz)
    nens_pg, nx, ny, trname, cname, crm_dt = CONFIGS[cfg_name][:6]
    nz, grid, xlen_cfg = CONFIGS[cfg_name][7:] or (60, "l60", None)
    tracers = {"none": idz.TRACERS_NONE, "kessler_shoc": idz.TRACERS_KESSLER_SHOC, "p3_shoc": idz.TRACERS_P3_SHOC}[trname]
    consts = {"default": idz.CONSTS_DEFAULT, "p3": idz.CONSTS_P3}[cname]
    zint = idz.l60_interfaces() if grid == "l60" else idz.uniform_interfaces(nz, 20000.0)
    xlen = xlen_cfg if xlen_cfg else nx * 1000.0
    ylen = ny * 1000.0 if ny > 1 else xlen
    names, pos, mass, idwv = idz.tracer_flags(tracers)
    lib, note = _oracle_lib()
    try:
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(threads)
    except Exception:
        pass
    f = idz.supercell_fields(nens, nx, ny, nz, zint, consts=consts, tracers=tracers, magnitude=0.1)
    if len(tracers) > 1:
        idz.add_tracer_blobs(f, tracers, xlen, ylen, zint)
    o = ao.OracleDycore(nens, nx, ny, nz, xlen, ylen, np.diff(zint), pos, mass, idwv, consts=consts, lib=lib)
    o.declare_current_profile_as_hydrostatic(f)
    o.time_step(copy.deepcopy(f), 0.2)      # warm-up (thread pool, page faults)
    t0 = time.time()
    ncyc, _ = o.time_step(f, crm_dt)
    el = time.time() - t0
    return {"value": nens * nz * ny * nx * ncyc / el, "seconds": el, "substeps": ncyc, "nens": nens, "threads": threads, "build": note,
            "grid": "%dx%dx%d" % (nx, ny, nz)}


def _spawn_cpu_workers(cfg_name, nens, threads, nproc):
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", "%s,%d,%d" % (cfg_name, nens, threads)],
                              env=env, stdout=subprocess.PIPE) for _ in range(nproc)]
    outs = []
    for pr in procs:
        o, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("cpu worker failed")
        outs.append(json.loads(o.decode().strip().splitlines()[-1]))
    return outs


def cpu_baseline(cfg_name):
    """the CPU path timed on THIS host on a bounded sample of the same workload -- (a) one core, (b) P
    processes x 1 core, each with its own members (how E3SM spreads CRMs over MPI ranks), (c) one process, OpenMP over the flux
    loop.  The program is the oracle (oracle/awfl_oracle.c, a C restatement of the reference algorithm: kind "port"); the
    reference itself needs YAKL, an absent submodule, and cannot be built here."""
    nens_pg, nx, ny = CONFIGS[cfg_name][:3]
    cores = _host_cores()
    _oracle_lib()                                              # build once, before the workers race for it
    n1 = 2 if ny > 1 else 32                                   # ~10 s of one core
    one = _spawn_cpu_workers(cfg_name, n1, 1, 1)[0]
    many = _spawn_cpu_workers(cfg_name, n1, 1, cores)
    nomp = 64 if ny > 1 else 1024
    omp = _spawn_cpu_workers(cfg_name, nomp, cores, 1)[0]
    agg = sum(m["value"] for m in many)
    model = _cpu_model()
    best = max(agg, omp["value"])
    return {"value": best, "unit": "cell-updates/s", "cores": cores, "kind": "port", "cpu_model": model,
            "one_core": {"value": one["value"], "cores": 1, "seconds": one["seconds"], "nens": n1},
            "processes_x_1_core": {"value": agg, "processes": cores, "cores": cores, "seconds_max": max(m["seconds"] for m in many),
                                   "nens_per_process": n1},
            "openmp": {"value": omp["value"], "threads": cores, "seconds": omp["seconds"], "nens": nomp},
            "sample_short": "oracle/awfl_oracle.c (C port of the reference algorithm; the reference needs YAKL, absent), %s grid, one timeStep = "
                            "%d sub-steps; value = best %d-core figure" % (one["grid"], one["substeps"], cores),
            "sample": "oracle/awfl_oracle.c (%s; a port of the reference algorithm: the reference needs YAKL, an absent submodule) on %s, "
                      "%s grid, one timeStep = %d sub-steps: 1 core x %d members %.1f s; %d processes x 1 core x %d members %.1f s; "
                      "OpenMP %d threads x %d members %.1f s; `value` = the better of the two %d-core figures"
                      % (one["build"], model, one["grid"], one["substeps"], n1, one["seconds"], cores, n1, max(m["seconds"] for m in many),
                         cores, nomp, omp["seconds"], cores)}


class Job:
    """One config resident on this rank's GPU: coupler + dycore + synthetic inputs."""

    def __init__(self, cfg_name, args, dev, rank, world, nens_override=0, perens=False):
        import numpy as np
        import torch
        from pam_amd import Dycore, PamCoupler, idealized as idz, parallel
        self.torch, self.parallel, self.idz = torch, parallel, idz
        nens_pg, nx, ny, trname, cname, crm_dt, desc = CONFIGS[cfg_name][:7]
        nz_cfg, grid_cfg, xlen_cfg = CONFIGS[cfg_name][7:] or (60, "l60", None)
        if nens_override > 0:
            nens_pg = nens_override
        self.nens_total = nens_pg * world
        if args.scaling == "strong":
            self.nens_total = nens_pg
            lo, hi = parallel.shard_range(nens_pg, rank, world)
            nens_pg = hi - lo
            if nens_pg < 1:
                raise SystemExit("bench.py: --scaling strong leaves rank %d without members" % rank)
        self.cfg_name, self.desc, self.crm_dt = cfg_name, desc, crm_dt
        self.nens, self.nx, self.ny, self.nz = nens_pg, nx, ny, nz_cfg
        self.tracers = {"none": idz.TRACERS_NONE, "kessler_shoc": idz.TRACERS_KESSLER_SHOC, "p3_shoc": idz.TRACERS_P3_SHOC}[trname]
        self.consts = {"default": idz.CONSTS_DEFAULT, "p3": idz.CONSTS_P3}[cname]
        self.nt = len(self.tracers)
        self.zint = idz.l60_interfaces() if grid_cfg == "l60" else idz.uniform_interfaces(nz_cfg, 20000.0)
        self.xlen = xlen_cfg if xlen_cfg else nx * 1000.0
        self.ylen = ny * 1000.0 if ny > 1 else self.xlen
        self.world = world
        # ---- coupler + dycore, inputs resident in HBM before anything is timed
        coupler = PamCoupler(dev)
        coupler.set_option("crm_dt", crm_dt)
        for k, v in self.consts.items():
            coupler.set_option(k, v)
        coupler.allocate_coupler_state(self.nz, ny, nx, nens_pg)
        self.perens = bool(perens or getattr(args, "perens", 0))
        if self.perens:
            # every member on its OWN vertical grid (the coupler's general contract: set_grid(..., realConst2d), pam_coupler.h:163-181 --
            # an MMF host puts every CRM under a different GCM column): the interior interfaces move by up to +-2 %, differently for
            # every member (bottom and top stay), so every (level, member) has its own WENO matrices (Dycore.h:897-940)
            a = 0.02 * (((np.arange(nens_pg) * 37) % 101) - 50.0) / 50.0
            kfrac = 1.0 - np.arange(self.nz + 1) / float(self.nz)
            coupler.set_grid(self.xlen, self.ylen, np.asarray(self.zint)[:, None] * (1.0 + a[None, :] * kfrac[:, None]))
        else:
            coupler.set_grid(self.xlen, self.ylen, self.zint)
        for n, p, m in self.tracers:
            coupler.add_tracer(n, "", p, m)
        dycore = Dycore()
        dycore.init(coupler)
        if getattr(args, "fold", "auto") != "auto":
            dycore.set_yz_fold(args.fold)
        if getattr(args, "tailfusion", "auto") != "auto":
            dycore.set_tail_fusion(args.tailfusion)
        if args.seg > 0:
            dycore.set_flux_segment(args.seg)
        if args.span >= 0:
            dycore.set_flux_span(args.span)
        if args.fused >= 0:
            dycore.set_fused_stage(args.fused)
        if args.lanes != "auto" or args.xkernels != "auto":
            dycore.set_lane_mapping(args.lanes, args.xkernels)
        if args.xtile:
            dycore.set_x_tile(*[int(v) for v in args.xtile.split(",")])
        if args.tilefusion != "auto":
            dycore.set_tile_fusion(args.tilefusion)
        if args.ftileparts != "auto":
            dycore.set_flux_tile_parts(args.ftileparts)
        if args.stateparts != "auto":
            dycore.set_tile_state_parts(args.stateparts)
        if args.ftile:
            ty, tz = [int(v) for v in args.ftile.split(",")]
            dycore.set_flux_tile("auto", ty, tz)
        if args.xexchange != "auto":
            dycore.set_x_exchange(args.xexchange)
        if args.trgroup != 0 or args.trprefetch:
            dycore.set_tracer_grouping(2 if (args.trprefetch and not args.trgroup) else args.trgroup, args.trprefetch)
        self.lane_mapping = dycore.get_lane_mapping()
        self.chunks = args.chunks
        self.lds_floor = args.lds_floor
        if args.chunks >= 0:
            dycore.set_ensemble_chunks(args.chunks, args.lds_floor)
        if args.indep >= 0:
            dycore.set_range_schedule(bool(args.indep))
        if args.graph != "auto":
            dycore.set_graph_replay(args.graph)
        if args.tuning:
            dycore.set_launch_tuning(*[int(v) for v in args.tuning.split(",")])
        nens_gen = min(16, nens_pg)
        f = idz.supercell_fields(nens_gen, nx, ny, self.nz, self.zint, consts=self.consts, tracers=self.tracers,
                                 magnitude=0.1, id0=rank * 1000)
        if self.nt > 1:
            idz.add_tracer_blobs(f, self.tracers, self.xlen, self.ylen, self.zint)
        if args.limiter:   # exact zeros in the vapour beside moist air: the FCT limiter acts on water_vapor in every stage
            idz.carve_dry_air(f, self.tracers, spread=args.limiter > 1)
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
        self.coupler, self.dycore = coupler, dycore

    def one_step(self):
        if self.world > 1 or getattr(self, "sharded", False):
            return self.parallel.sharded_time_step(self.dycore, self.coupler)
        return self.dycore.timeStep(self.coupler)

    def timed(self, steps, warmup, dist=None, backend="nccl", dev=None):
        torch = self.torch
        for _ in range(warmup):
            self.one_step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        substeps = 0
        for _ in range(steps):
            substeps += self.one_step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        updates = float(self.nens * self.nz * self.ny * self.nx * substeps)
        self.rank_elapsed = (elapsed, elapsed)
        if dist is not None:
            t = torch.tensor([elapsed, -elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, emin = float(t[0].item()), -float(t[1].item())
            self.rank_elapsed = (emin, elapsed)     # fastest and slowest rank: load imbalance is visible in the line
            u = torch.tensor([updates], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(u, op=dist.ReduceOp.SUM)
            updates = float(u.item())
        return updates, elapsed, substeps

    def kernel_pass(self):
        d = self.dycore
        d.reset_kernel_timing()
        self.one_step()
        self.torch.cuda.synchronize()
        out = {}
        for name in STAGE_KERNELS + ("init_prim", "finalize", "cfl", "flux_xy", "flux_z"):
            ms, n = d.get_kernel_timing(name)
            if n:
                out[name] = {"launches": n, "avg_ms": ms / n, "total_ms": ms}
        return out

    def close(self):
        self.dycore.finalize(self.coupler)
        self.coupler.dm.finalize()
        self.torch.cuda.empty_cache()


def stage_count(timings):
    """tendency stages in a timed pass: every stage has exactly one update-type launch"""
    k = "xupd" if "xupd" in timings else "update"
    return max(1, timings[k]["launches"])


def stage_rooflines(job, alone):
    """Per-kernel accounting of one tendency stage (nothing co-running).  `own_bytes`: what the kernel must itself move given
    the kernel split -- every field it reads or writes, once (sub-step-start values are read in 2 of the 3 stages: x 2/3);
    `flops`: FP64 operations of its WENO polynomials (~183 per polynomial: 119/128 FP64 instructions, half of them FMAs -- ISA
    count), ~60 per cell and variable for the update arithmetic and ~250 for a pow."""
    cells = float(job.nens * job.nz * job.ny * job.nx)
    nt, d3 = job.nt, job.ny > 1
    fb = cells * 8.0                       # one interior-sized field
    pb = fb * (job.nz + 6) / job.nz        # one prim field (3 ghost levels below and above)
    fused = "xupd" in alone
    poly = 183.0 * 1.03
    nall = 3 if d3 else 2                  # sweep directions
    acct = {}
    if fused:
        nyz = nall - 1                     # the flux kernel sweeps y (3-D only) and z: mass + tracers as faces, the rest as differences
        acct["flux"] = (nyz * (6 + nt) * pb + nyz * (5 + nt) * fb, cells * nyz * (6 + nt) * poly)
        # fused x-sweep: the state and water vapour complete (stage input 7 fields, sub-step start 6, y/z flux differences 5 and
        # vapour faces 1 per direction, FCT seed; writes rho, u, v, w, theta, rho*theta, vapour, its seed, its x flux (+ the face
        # mass flux when further tracers follow))
        state = (7 * pb + (2.0 / 3.0) * 6 * pb + nyz * 6 * fb + fb + 7 * pb + 2 * fb + (fb if nt > 1 else 0), cells * (7 * poly + 6 * 60.0 + 40.0))
        # sweeps of the further tracers.  Phase 1 (FCT multipliers): per tracer the stage input, its y/z faces and seed, per pair
        # the face mass flux; writes the multiplier of EVERY cell (own_multiplier_cell<DENSE>: a complete field).  Phase 2 (complete
        # update): per tracer the stage input, the sub-step start, its y/z faces and the multipliers of its own line (those of the
        # y/z neighbours are re-reads of other lines' values: L2-shared, not counted), per pair the mass flux and the three
        # densities; writes the new value and seed
        ntr, npair = nt - 1, nt // 2
        ph1 = (ntr * (pb + nyz * fb + fb + fb) + npair * fb, cells * ntr * (poly + 20.0))
        ph2 = (ntr * (pb + (2.0 / 3.0) * pb + nyz * fb + fb + pb + fb) + npair * (fb + (2 + 2.0 / 3.0) * pb), cells * ntr * (poly + 60.0))
        if "xtr1" in alone:
            acct["xupd"], acct["xtr1"] = state, ph1
        else:
            acct["xupd"] = (state[0] + ph1[0], state[1] + ph1[1])
        acct["xtr2"] = ph2
        acct["ptail"] = (2 * fb, cells * 250.0)
        acct["trfix"] = (0.0, 0.0)         # only where the limiter acted
    else:
        acct["flux"] = (nall * (6 + nt) * pb + nall * (5 + nt) * fb, cells * nall * (6 + nt) * poly)
        acct["fct_mult"] = (nt * (nall + 2) * fb, cells * nt * 20.0)
        acct["update"] = (nall * (5 + nt) * fb + (6 + nt) * pb * (1 + 2.0 / 3.0) + nt * fb + (6 + nt) * pb + nt * fb,
                          cells * ((5 + nt) * 60.0 + 250.0))
    out = []
    nstage = stage_count(alone)
    for name, (nbytes, flops) in acct.items():
        if name not in alone:
            continue
        s = alone[name]["total_ms"] / nstage * 1e-3
        kname = "awfl_%s_kernel" % name.replace("fct_mult", "fct")
        if name.startswith("xtr"):
            kname = "awfl_xtr_kernel<%s>" % name[3:]
        out.append({"kernel": kname, "ms_per_stage": alone[name]["total_ms"] / nstage,
                    "launches_per_stage": alone[name]["launches"] / nstage,
                    "own_bytes_per_stage": nbytes, "own_GBps": nbytes / s / 1e9, "hbm_frac": nbytes / s / 1e9 / HBM_PEAK_GBS,
                    "fp64_flops_per_stage": flops, "fp64_TFLOPs": flops / s / 1e12,
                    "valu_frac": flops / s / 1e12 / FP64_VALU_PEAK_TFLOPS})
    return out


PROFILE_ROUND = "r06"


def load_profile(cfg_name):
    """per-kernel PMC figures of this config (tools/profile_c2.sh / profile_small.sh -> profiles/<round>_<cfg>_traffic.json), valid only
    for the build they were taken from: content hash of pam_amd/csrc"""
    path = os.path.join(ROOT, "profiles", "%s_%s_traffic.json" % (PROFILE_ROUND, cfg_name))
    if not os.path.exists(path):
        return None, "no PMC profile for this configuration (%s)" % os.path.basename(path)
    prof = json.load(open(path))
    if prof.get("csrc_hash") != csrc_hash():
        return None, "%s was measured on another build of pam_amd/csrc: not reported" % os.path.basename(path)
    # the launch shapes of the tile kernels follow the device's CU count (xtile_geometry, choose_flux_tiles): a profile taken on a part
    # with another CU count describes other launches
    try:
        import torch
        ncu = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    except Exception:
        ncu = None
    if prof.get("compute_units") and ncu and prof["compute_units"] != ncu:
        return None, "%s was measured on a part with %d compute units (this one: %d): not reported" % (os.path.basename(path), prof["compute_units"], ncu)
    return prof, "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE / SQ_INSTS_VALU / GRBM_GUI_ACTIVE, separate passes, this build (%s)" % prof["csrc_hash"]


def measure_roofline(job, args, profile_key=None):
    """HIP-event pass over the stage kernels of `job` + the roofline object of its dominant kernel (see the module docstring)."""
    d = job.dycore
    cells = job.nens * job.nz * job.ny * job.nx
    nt = job.nt
    d.set_kernel_timing(True)
    kernels = job.kernel_pass()                    # shipped configuration (chunks overlap: durations include contention)
    d.set_ensemble_chunks(1)                       # every stage kernel on its own: whole ensemble per launch
    alone = job.kernel_pass()
    d.set_ensemble_chunks(args.chunks if args.chunks >= 0 else 0, args.lds_floor)
    d.set_kernel_timing(False)
    kernel_rooflines = stage_rooflines(job, alone)
    stage = [k for k in STAGE_KERNELS if k in alone]
    if not stage:
        return None, kernels, kernel_rooflines
    nstage = stage_count(alone)
    dom = max(stage, key=lambda k: alone[k]["total_ms"])
    avg_s = alone[dom]["total_ms"] / nstage * 1e-3    # per stage (the y and z sweeps may be two launches of one kernel)
    alg_bytes = cells / 3.0 * 64.0 * (5 + nt)            # 64*(5+NT) B per cell-update, 1/3 per stage
    achieved = alg_bytes / avg_s / 1e9
    kname = "awfl_%s_kernel" % dom.replace("fct_mult", "fct")
    if dom.startswith("xtr"):
        kname = "awfl_xtr_kernel<%s>" % dom[3:]
    mine = [k for k in kernel_rooflines if k["kernel"] == kname]
    prof, tnote = (load_profile(profile_key) if profile_key else (None, "not a profiled workload"))
    traffic, stage_traffic, valu, stage_valu_insts = None, None, None, 0.0
    if prof is not None:
        pk = prof["kernels"]

        def pkey(name):      # "awfl_xtr_kernel<2>" is one template family in the profile; small ensembles run the _tile_ forms
            for cand in (name, name.replace("xtr_kernel", "xtrn_kernel"), name.replace("_kernel", "_tile_kernel"), name.split("<")[0],
                         name.split("<")[0].replace("_kernel", "_tile_kernel")):
                if cand in pk:
                    return cand
            if name.startswith("awfl_trfix") and "awfl_trfix_flat_kernel" in pk:
                return "awfl_trfix_flat_kernel"
            return name
        def pkeys(name):     # every kernel family of the profile that bench.py's `name` stands for: with per-member vertical grids
            ks = [pkey(name)]    # the "flux" launches are awfl_flux_kernel (y) + awfl_fluxz_pe_kernel (z)
            if name == "awfl_flux_kernel" and "awfl_fluxz_pe_kernel" in pk:
                ks.append("awfl_fluxz_pe_kernel")
            return [k for k in ks if k in pk]

        def psum(name, field):
            vals = [pk[k][field] for k in pkeys(name) if field in pk[k]]
            return sum(vals) if vals else None
        traffic = psum(kname, "hbm_bytes_per_stage")
        for kr in kernel_rooflines:
            tr_ = psum(kr["kernel"], "hbm_bytes_per_stage")
            if tr_ is not None:
                kr["traffic"] = tr_
        names = set(k for kr in kernel_rooflines for k in pkeys(kr["kernel"]))
        stage_traffic = sum(pk[k]["hbm_bytes_per_stage"] for k in names)
        stage_valu_insts = sum(pk[k].get("valu_insts_per_stage", 0.0) for k in names)
        c = {"valu_insts_per_stage": psum(kname, "valu_insts_per_stage"), "busy_cycles_per_xcd_per_stage": psum(kname, "busy_cycles_per_xcd_per_stage")}
        c = {k: v for k, v in c.items() if v}
        if "valu_insts_per_stage" in c and "busy_cycles_per_xcd_per_stage" in c:
            # counter-based VALU figures of the dominant kernel: a wave-level fp64 VALU instruction occupies its SIMD for 4
            # cycles; 1024 SIMDs.  issue_frac: of the cycles the chip was busy (at the clock it actually held, ~1.9 GHz under this
            # FP64 load); frac: of the 2.4 GHz spec, with this run's duration
            simd_cycles = c["valu_insts_per_stage"] * 4.0 / 1024.0
            valu = {"bound": "fp64-valu", "unit": "SIMD issue cycles per stage", "achieved": simd_cycles,
                    "issue_frac_at_sustained_clock": simd_cycles / c["busy_cycles_per_xcd_per_stage"],
                    "sustained_clock_GHz_approx": c["busy_cycles_per_xcd_per_stage"] / avg_s / 1e9,   # profile's busy cycles / this run's duration
                    "peak": avg_s * 2.4e9, "frac": simd_cycles / (avg_s * 2.4e9),
                    "source": "SQ_INSTS_VALU and GRBM_GUI_ACTIVE of the profile, duration of this run"}
    if valu is None and mine:
        valu = {"bound": "fp64-valu", "achieved": mine[0]["fp64_TFLOPs"], "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": mine[0]["valu_frac"], "source": "instruction-count model (no counter profile of this build)"}
    stage_ms = sum(alone[k]["total_ms"] for k in stage) / nstage
    # the whole stage against its FP64-issue floor at the clock the chip held (every wave-level VALU instruction of every stage kernel
    # occupies its SIMD for 4 cycles; 1024 SIMDs): the roofline that binds this path
    stage_valu_frac = None
    if prof is not None and valu and "sustained_clock_GHz_approx" in valu and stage_valu_insts:
        stage_valu_frac = stage_valu_insts * 4.0 / 1024.0 / (valu["sustained_clock_GHz_approx"] * 1e9) / (stage_ms * 1e-3)
    # which roofline binds the dominant kernel: FP64 vector issue when its VALU fraction exceeds its HBM fraction
    hbm_frac = achieved / HBM_PEAK_GBS
    bound = "fp64-valu" if (valu and valu["frac"] > hbm_frac) else "hbm"
    roofline = {"bound": bound, "kernel": kname, "achieved": achieved, "peak": HBM_PEAK_GBS,
                "frac_definition": "frac = algorithmic bytes of a WHOLE tendency stage (SURVEY 8d: 64*(5+NT) B per cell-update / 3) divided by the "
                                   "time of the stage's dominant kernel ALONE, over the 8 TB/s HBM peak: it charges one kernel with the whole "
                                   "stage's bytes and so flatters the stage.  stage_frac = the same bytes over ALL stage kernels back to back -- "
                                   "the figure to quote for the stage; value * 64*(5+NT) B / peak is the whole step",
                "unit": "GB/s", "frac": hbm_frac, "hbm_frac": hbm_frac, "traffic": traffic, "traffic_note": tnote,
                "ms_per_stage": alone[dom]["total_ms"] / nstage, "launches_per_stage": alone[dom]["launches"] / nstage,
                "alg_bytes_per_launch": alg_bytes,
                "note": "achieved/peak/frac are the HBM view asked for by the metric (SURVEY 8d algorithmic bytes of a whole stage / "
                        "this kernel's time per stage, measured with one ensemble range, nothing co-running); `bound` names the "
                        "roofline that actually limits the kernel, `valu` quantifies it; all stage kernels back to back take "
                        "%.2f ms" % stage_ms,
                "stage_ms_back_to_back": stage_ms,
                "stage_frac": alg_bytes / (stage_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "stage_traffic": stage_traffic,
                "stage_traffic_ratio": (stage_traffic / alg_bytes) if stage_traffic else None,
                "valu": valu,
                # flat copies of the figures that bind
                "valu_frac_spec_clock": valu.get("frac") if valu else None,
                "valu_issue_frac": valu.get("issue_frac_at_sustained_clock") if valu else None,
                "sustained_clock_GHz": valu.get("sustained_clock_GHz_approx") if valu else None,
                "stage_valu_frac": stage_valu_frac}
    return roofline, kernels, kernel_rooflines


def modules_timing(torch, dev):
    """the coupler modules around the dycore at the C2 grid (1024 x 32x32x60), HIP events around the C-ABI calls:
    ms per call and the bytes each must move (fields read + written, once)."""
    from pam_amd import PamCoupler, Microphysics, modules, idealized as idz
    nens, nx, ny, nz = 1024, 32, 32, 60
    zint = idz.l60_interfaces()
    cells = nens * nx * ny * nz
    out = {"grid": "1024 x 32x32x60 (C2)", "unit": "ms per call"}

    def timeit(fn, n=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    f = idz.supercell_fields(16, nx, ny, nz, zint, tracers=(("water_vapor", True, True),), magnitude=0.5)
    c = PamCoupler(dev)
    c.set_option("crm_dt", 2.0)
    c.set_option("gcm_physics_dt", 1200.0)
    c.allocate_coupler_state(nz, ny, nx, nens)
    c.set_grid(nx * 1000.0, ny * 1000.0, zint)
    micro = Microphysics()
    micro.init(c)
    dm = c.get_data_manager_device_readwrite()
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp"):
        dm.get(k).copy_(torch.from_numpy(f[k]).to(dev).repeat(1, 1, 1, nens // 16))
    dm.get("water_vapor").copy_(torch.from_numpy(f["tracers"][0]).to(dev).repeat(1, 1, 1, nens // 16) * 1.3)
    dm.get("precip_liquid").copy_(dm.get("density_dry") * 1e-3)
    nsplit = micro.timeStep(c)
    t = timeit(lambda: micro.timeStep(c))
    # one sub-cycle: the limit kernel reads rho_r and rho_d; the column kernel reads rho_d, rho_v, rho_c, rho_r, T and writes the
    # last four.  Each further sub-cycle: 5 reads + 4 writes, and the Exner function written once and read per cycle.
    kb = cells * 8.0 * (11 if nsplit == 1 else 2 + 10 * nsplit)
    out["kessler_time_step"] = {"ms": t, "bytes": kb, "GBps": kb / t / 1e6, "hbm_frac": kb / t / 1e6 / HBM_PEAK_GBS, "rainsplit": nsplit,
                                "note": "limit kernel 2 reads, column kernel 5 reads + 4 writes per cell (rounds 4-5: 26 passes, the "
                                        "conversions in a kernel of their own)"}
    # the same call on a rain-free, cloud-free state (most columns of a real run): wavefronts without rain skip the powers of rain amounts
    dm.get("water_vapor").copy_(torch.from_numpy(f["tracers"][0]).to(dev).repeat(1, 1, 1, nens // 16) * 0.5)
    dm.get("precip_liquid").zero_()
    dm.get("cloud_liquid").zero_()
    micro.timeStep(c)
    t = timeit(lambda: micro.timeStep(c))
    out["kessler_time_step_no_rain"] = {"ms": t, "bytes": cells * 8.0 * 11, "GBps": cells * 8.0 * 11 / t / 1e6,
                                        "hbm_frac": cells * 8.0 * 11 / t / 1e6 / HBM_PEAK_GBS}
    t = timeit(lambda: modules.sponge_layer(c))
    nsp = 5                   # sponge_layer.h:8-95: the top 5 of 60 levels
    nfld = 5 + len(c.get_tracer_names())          # rho_d, u, v, w, T + every tracer (sponge_layer.h:54-62); w's mean is zero: not read
    sp_bytes = nens * nx * ny * nsp * (3 * nfld - 1) * 8.0
    out["sponge_layer"] = {"ms": t, "bytes": sp_bytes, "GBps": sp_bytes / t / 1e6, "hbm_frac": sp_bytes / t / 1e6 / HBM_PEAK_GBS,
                           "note": "%d fields, top 5 levels: read for the mean (not w), read + written for the relaxation; the second read "
                                   "comes out of the caches, so the compulsory HBM traffic is 2/3 of this" % nfld}
    del micro, dm, c
    torch.cuda.empty_cache()
    c = PamCoupler(dev)
    c.set_option("crm_dt", 2.0)
    c.set_option("gcm_physics_dt", 1200.0)
    c.allocate_coupler_state(nz, ny, nx, nens)
    c.set_grid(nx * 1000.0, ny * 1000.0, zint)
    for n in ("water_vapor", "cloud_water", "ice", "cloud_water_num", "ice_num", "rain_num"):
        c.add_tracer(n, "", True, n in ("water_vapor", "cloud_water", "ice"))
    dm = c.get_data_manager_device_readwrite()
    for k in ("density_dry", "uvel", "vvel", "wvel", "temp"):
        dm.get(k).copy_(torch.from_numpy(f[k]).to(dev).repeat(1, 1, 1, nens // 16))
    dm.get("water_vapor").copy_(torch.from_numpy(f["tracers"][0]).to(dev).repeat(1, 1, 1, nens // 16))
    for k in ("gcm_density_dry", "gcm_temp", "gcm_water_vapor"):
        src = {"gcm_density_dry": "density_dry", "gcm_temp": "temp", "gcm_water_vapor": "water_vapor"}[k]
        dm.get(k).copy_(dm.get(src).mean(dim=(1, 2)) * 1.01)
    modules.compute_gcm_forcing_tendencies(c)
    t = timeit(lambda: modules.compute_gcm_forcing_tendencies(c), 3)
    out["compute_gcm_forcing_tendencies"] = {"ms": t, "bytes": cells * 10 * 8.0, "GBps": cells * 10 * 8.0 / t / 1e6,
                                             "hbm_frac": cells * 10 * 8.0 / t / 1e6 / HBM_PEAK_GBS,
                                             "note": "column means of rho_d, u, v, T, rho_v, rho_l, rho_i, nc, ni, nr (gcm_forcing.h:149-174): "
                                                     "10 fields read (rounds 1-5 counted 5 of them)"}
    t = timeit(lambda: modules.apply_gcm_forcing_tendencies(c), 3)
    out["apply_gcm_forcing_tendencies"] = {"ms": t, "bytes": cells * 20 * 8.0, "GBps": cells * 20 * 8.0 / t / 1e6,
                                           "hbm_frac": cells * 20 * 8.0 / t / 1e6 / HBM_PEAK_GBS,
                                           "note": "the same 10 fields read and written in place (gcm_forcing.h:361-429; the hole-filling sums "
                                                   "ride along; rounds 1-5 counted 5 + 5)"}
    del dm, c
    torch.cuda.empty_cache()
    return out


LINE_LIMIT = 4096              # the driver parses the LAST stdout line; round 4's 32 KB object was not parsed


def _sig(x, n=6):
    """numbers of the compact line carry six significant digits (value and ms_per_step are kept exact)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float("%.*g" % (n, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


def compact_line(full):
    """The line the driver parses: the contract's fields, `roofline` and `cpu_baseline` as flat objects of numbers, the other
    configurations as one number each.  Everything else (per-kernel tables, per-config rooflines, module timings, the prose that
    defines each figure) is in bench_detail.json beside this file and on stderr."""
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data")}
    c = full["config"]
    keep = ("scaling", "nens_total", "nens_per_gpu", "workload", "nx", "ny", "nz", "num_tracers", "crm_dt", "substeps_per_step",
            "parallelism", "collective", "ranks_seen", "launcher", "devices_seen", "limiter_input", "fct_rows_flagged_last_stage",
            "fct_rows", "rank_ms_per_step")
    out["config"] = _sig({k: c[k] for k in keep if k in c})
    m = c.get("lane_mapping")
    if m:
        out["config"]["lanes"] = ("flat" if m.get("yz_flat") else "member") + "+" + (("xtile-shfl" if m.get("x_shuffles") else "xtile") if m.get("x_tiles") else "xsweep")
    r = full.get("roofline")
    if r:
        rr = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "ms_per_stage", "alg_bytes_per_launch",
                                    "stage_ms_back_to_back", "stage_frac", "stage_traffic", "stage_traffic_ratio", "valu_frac_spec_clock",
                                    "valu_issue_frac", "sustained_clock_GHz", "stage_valu_frac")}
        v = r.get("valu")
        if v:
            rr["valu"] = {k: v[k] for k in ("frac", "issue_frac_at_sustained_clock", "sustained_clock_GHz_approx") if k in v}
            rr["valu"]["source"] = "pmc" if "issue_frac_at_sustained_clock" in v else "model"
        out["roofline"] = _sig(rr)
    else:
        out["roofline"] = None
    b = full.get("cpu_baseline")
    if b:
        bb = {k: b.get(k) for k in ("value", "unit", "cores", "kind", "cpu_model")}
        for k in ("one_core", "processes_x_1_core", "openmp"):
            if isinstance(b.get(k), dict):
                bb[k] = {"value": b[k]["value"], "s": b[k].get("seconds", b[k].get("seconds_max")),
                         "nens": b[k].get("nens", b[k].get("nens_per_process"))}
        bb["sample"] = (b.get("sample_short") or b.get("sample") or "")[:200]
        out["cpu_baseline"] = _sig(bb)
    else:
        out["cpu_baseline"] = None
    o = full.get("other_configs")
    if o:
        oo = {}
        for k, v in o.items():
            if k == "modules":
                oo["modules_ms"] = {kk: vv["ms"] for kk, vv in v.items() if isinstance(vv, dict) and "ms" in vv}
                oo["modules_hbm_frac"] = {kk: vv["hbm_frac"] for kk, vv in v.items() if isinstance(vv, dict) and "hbm_frac" in vv}
            elif isinstance(v, dict):
                oo[k] = v.get("value")
                rf = v.get("roofline") or {}
                if k in ("c3", "c4"):
                    oo[k + "_ms_per_step"] = v.get("ms_per_step")
                if k in ("c3", "c4", "c2_perens", "c4_perens", "ref_nens1", "c2grid_nens1") and rf:
                    oo[k + "_stage_frac"] = rf.get("stage_frac")
                    oo[k + "_stage_traffic_ratio"] = rf.get("stage_traffic_ratio")
        oo["unit"] = "cell-updates/s"
        out["other"] = _sig(oo, 4)
    out["detail"] = "bench_detail.json"
    return out


def emit(full, args):
    """full object -> bench_detail.json + stderr; compact object (< LINE_LIMIT bytes, strict JSON) -> the last stdout line"""
    detail = json.dumps(_sig(full, 17), allow_nan=False)      # strict JSON: a NaN / inf becomes null
    try:
        with open(os.path.join(ROOT, args.detail), "w") as fh:
            fh.write(detail + "\n")
    except OSError as e:
        sys.stderr.write("bench.py: cannot write %s: %r\n" % (args.detail, e))
    sys.stderr.write("bench.py detail: " + detail + "\n")
    sys.stderr.flush()
    line = json.dumps(compact_line(full), allow_nan=False, separators=(",", ":"))
    if len(line) >= LINE_LIMIT:            # never print a line the driver cannot parse: drop the optional blocks, largest first
        c = compact_line(full)
        for k in ("other", "detail"):
            c.pop(k, None)
            line = json.dumps(c, allow_nan=False, separators=(",", ":"))
            if len(line) < LINE_LIMIT:
                break
    assert len(line) < LINE_LIMIT, "bench line of %d bytes" % len(line)
    sys.stdout.write(line + "\n")
    sys.stdout.flush()


def worker(args):
    import torch
    import torch.distributed as dist
    from pam_amd import idealized as idz

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (python -m torch.distributed.run "
                         "--nproc-per-node N bench.py --gpus N), or run `python bench.py --gpus N` without WORLD_SIZE set and "
                         "let it start the ranks" % (args.gpus, world))
    if os.environ.get("PAM_AMD_BENCH_DRYRUN") == "1":
        # launcher rehearsal without a GPU (tests/test_bench_launcher.py): rendezvous over gloo, count the ranks, no dycore
        from pam_amd import parallel
        nens_cfg = args.nens if args.nens > 0 else CONFIGS[args.config][0]
        if args.scaling == "strong":
            lo, hi = parallel.shard_range(nens_cfg, rank, world)
            mine = hi - lo
        else:
            mine = nens_cfg
        n, shards = 1.0, [mine]
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            t = torch.ones(1, dtype=torch.float64)
            dist.all_reduce(t)
            n = float(t.item())
            g = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(g, torch.tensor([mine], dtype=torch.int64))
            shards = [int(x.item()) for x in g]
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": n, "scaling": args.scaling, "config": args.config,
                              "shard_sizes": shards, "nens_total": sum(shards)}))
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    # PAM_AMD_DIST_BACKEND=gloo is the rehearsal switch: several ranks on ONE GPU (RCCL refuses that), collectives on CPU.
    # It is chosen automatically when the box has fewer GPUs than ranks.
    backend = os.environ.get("PAM_AMD_DIST_BACKEND", "nccl" if ndev >= world else "gloo")
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    # PAM_AMD_DIST_SELFTEST=1: run the N>1 code path (process group, RCCL all-reduce of dt, barrier) with a single rank --
    # the only way to exercise the RCCL calls on a 1-GPU box
    selftest = world == 1 and os.environ.get("PAM_AMD_DIST_SELFTEST") == "1"
    if world > 1 or selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dd = dist if (world > 1 or selftest) else None
    ranks_seen = 1
    if dd is not None:      # every rank adds one through the same backend the dt exchange uses: N ranks must be N
        t = torch.ones(1, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dd.all_reduce(t)
        ranks_seen = int(round(float(t.item())))

    job = Job(args.config, args, dev, rank, world, args.nens)
    job.sharded = dd is not None
    updates, elapsed, substeps = job.timed(args.steps, args.warmup, dd, backend, dev)
    value = updates / elapsed
    fct_rows = job.dycore.debug_fct_rows()      # rows the LAST stage's limiter flagged / all rows
    cells = job.nens * job.nz * job.ny * job.nx
    nt = job.nt

    # ---- per-kernel durations (HIP events on the stream each kernel is launched on), separate un-timed passes
    roofline, kernels, kernel_rooflines = None, {}, []
    if not args.no_kernel_timing:
        plain = args.scaling == "weak" and not args.limiter
        pkey = args.config if (plain and args.nens == 0) else ("c2grid_nens%d" % args.nens if (plain and args.config == "c2") else None)
        if plain and args.config == "ref" and args.nens in (0, 1):
            pkey = "ref_nens1"
        if args.perens:
            pkey = (args.config + "_perens") if (plain and args.nens == 0) else None
        roofline, kernels, kernel_rooflines = measure_roofline(job, args, pkey)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(args.config)
        except Exception as e:   # the baseline is a reported extra; never fail the bench line for it
            cpu = {"value": None, "unit": "cell-updates/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}

    rank_ms = [e / args.steps * 1e3 for e in job.rank_elapsed]
    desc, nens_pg, nens_total = job.desc % job.nens, job.nens, job.nens_total
    lane_mapping = job.lane_mapping
    ny, nx, nz, crm_dt = job.ny, job.nx, job.nz, job.crm_dt
    job.close()
    del job

    # ---- the other single-GPU configurations, same procedure, fewer steps (default N=1 run only)
    others = None
    if world == 1 and args.config == "c2" and args.nens == 0 and not args.no_other_configs and not args.limiter:
        others = {}

        def run_other(key, cfg, nens=0, limiter=0, steps=10, note=None, profile=None, warmup=1, perens=False):
            try:
                a2 = copy.copy(args)
                a2.limiter = limiter
                j = Job(cfg, a2, dev, 0, 1, nens, perens=perens)
                u, el, sub = j.timed(steps, warmup)
                rows = j.dycore.debug_fct_rows()
                others[key] = {"value": u / el, "unit": "cell-updates/s", "ms_per_step": el / steps * 1e3, "workload": j.desc % j.nens,
                               "num_tracers": j.nt, "substeps_per_step": sub / float(steps),
                               "hbm_frac": u / el * 64.0 * (5 + j.nt) / 1e9 / HBM_PEAK_GBS,
                               "fct_rows_flagged_last_stage": rows[0], "fct_rows": rows[1]}
                if note:
                    others[key]["note"] = note
                if not args.no_kernel_timing:
                    rf, _, krf = measure_roofline(j, a2, profile)
                    others[key]["roofline"] = rf
                    others[key]["kernel_rooflines"] = krf
                j.close()
                del j
            except Exception as e:
                others[key] = {"value": None, "error": repr(e)}
        run_other("c3", "c3", profile="c3")
        # (one GPU's shard of C4 takes 10-12 ms per step: 30 steps behind 3 warm-up steps -- at 10 + 1 the per-member variant came out
        #  15 % low in 2 of ~12 runs of round 6)
        run_other("c4", "c4", profile="c4", steps=30, warmup=3)
        # per-member vertical grids (every (level, member) has its own WENO matrices: Dycore.h:897-940, pam_coupler.h:163-181)
        run_other("c2_perens", "c2", perens=True, steps=5, profile="c2_perens",
                  note="C2 with every member on its own vertical grid: the z sweep stages each level's tables in LDS per workgroup (awfl_fluxz_pe_kernel)")
        run_other("c4_perens", "c4", perens=True, profile="c4_perens", steps=30, warmup=3, note="C4 (one GPU's shard) with per-member vertical grids")
        # the N = 1 denominators of the two strong-scaling rows and the per-GPU workload of C2 over 8 GPUs
        run_other("c4_full", "c4", nens=4096, steps=10,
                  note="BASELINE config C4 whole (nens = 4096, NT = 10) on ONE GPU: what c4 (one GPU's 512-member shard) is 1/8 of")
        run_other("c2_shard128", "c2", nens=128,
                  note="what one GPU runs of C2 strong-scaled over 8 GPUs (1024 / 8 members)")
        # the limiter acting on water vapour itself (NT = 1): the flagged paths of the state pass and the fix-up pass
        run_other("c2_limiter1", "c2", limiter=1, steps=5,
                  note="C2 with dry slabs in the vapour at the same place in every member (--limiter 1)")
        run_other("c2_limiter2", "c2", limiter=2, steps=5,
                  note="C2 with dry slabs at member-dependent places: nearly every row of 64 members flagged (--limiter 2)")
        # small ensembles (flat lanes + tile kernels): the reference's own input shape and the C2 grid with one member
        # (a timeStep of these is ~0.3 / ~0.8 ms: 40 steps behind 5 warm-up steps, or the first launches of a cold stream dominate)
        run_other("ref_nens1", "ref", steps=40, warmup=5, profile="ref_nens1",
                  note="the shape of the reference's input file (input_pama.yaml: 250x1, nens = 1, 50 levels), Kessler + SHOC tracers")
        run_other("c2grid_nens1", "c2", nens=1, steps=40, warmup=5, profile="c2grid_nens1", note="C2's 32x32x60 grid with ONE member")
        try:
            others["modules"] = modules_timing(torch, dev)
        except Exception as e:
            others["modules"] = {"error": repr(e)}

    if rank == 0:
        out = {"metric": "cell-updates/sec (AWFL dycore step)", "value": value, "unit": "cell-updates/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"scaling": args.scaling, "nens_total": nens_total, "nens_per_gpu": nens_pg, "workload": desc, "nx": nx, "ny": ny,
                          "nz": nz, "num_tracers": nt, "crm_dt": crm_dt, "substeps_per_step": substeps / args.steps,
                          "parallelism": "nens-shard x%d" % world,
                          "limiter_input": args.limiter, "fct_rows_flagged_last_stage": fct_rows[0], "fct_rows": fct_rows[1],
                          "collective": None if world == 1 else "all-reduce(MIN) of dt, 8 B per timeStep, backend %s%s" % (
                              backend, "" if ndev >= world else " (rehearsal: %d ranks share %d GPU)" % (world, ndev)),
                          "ranks_seen": ranks_seen, "device": str(dev), "lane_mapping": lane_mapping,
                          "rank_ms_per_step": {"min": rank_ms[0], "max": rank_ms[1]}},
               "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels, "kernel_rooflines": kernel_rooflines,
               "other_configs": others}
        emit(out, args)
    if dd is not None:
        dist.destroy_process_group()


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"))
    ap.add_argument("--nens", type=int, default=0, help="members per GPU (weak) / in total (strong); default: the config's")
    ap.add_argument("--seg", type=int, default=0, help="flux-kernel chunk length (default: library default)")
    ap.add_argument("--span", type=int, default=-1, help="flux-kernel faces per thread (default: automatic)")
    ap.add_argument("--chunks", type=int, default=-1, help="internal ensemble chunks / HIP streams (default: automatic)")
    ap.add_argument("--fused", type=int, default=-1, help="1/0: fused x-sweep stage / three-kernel stage (default: library default)")
    ap.add_argument("--lds-floor", type=int, default=0, help="tuning: LDS request per flux workgroup when chunks > 1 (residency cap)")
    ap.add_argument("--limiter", type=int, default=0,
                    help="1: input with dry slabs in the water vapour (same place in every member), so that the FCT limiter acts on "
                         "water_vapor itself in every stage along the slab edges (with one tracer: times the flagged path of the NT=1 "
                         "tail); 2: slabs at member-dependent places (nearly every row of 64 members flagged: worst case of the "
                         "sparse multiplier); default 0: smooth vapour")
    ap.add_argument("--lanes", default="auto", choices=("auto", "member", "flat"),
                    help="lanes of the y/z sweeps: 64 members of one line / 64 items of the flattened (x, member) axis (auto: flat when nens < 64)")
    ap.add_argument("--xkernels", default="auto", choices=("auto", "sweep", "tile"),
                    help="x direction: a wavefront per line span / a lane per cell with LDS exchange (auto: tile when nens < 64)")
    ap.add_argument("--xtile", default="", help="tile geometry W,tc,lpb (0 = automatic each)")
    ap.add_argument("--trgroup", type=int, default=0, choices=(0, 1, 2, 4), help="further tracers per wavefront of the separate x tracer sweeps (0 = automatic)")
    ap.add_argument("--trprefetch", type=int, default=0, choices=(0, 1), help="phase 2 of those sweeps: next trip's loads one trip ahead")
    ap.add_argument("--tilefusion", default="auto", choices=("auto", "separate", "inside", "beside"),
                    help="x tile kernels of small ensembles: pressure pass / tracer phase 1 in launches of their own, inside the state kernel, "
                         "or inside its launch with phase 1 in workgroups beside the state pass")
    ap.add_argument("--stateparts", default="auto", choices=("auto", "one", "parts"), help="fused x tile kernel: the state pass by one lane / in three parts beside each other")
    ap.add_argument("--ftileparts", default="auto", choices=("auto", "behind", "beside"), help="y/z flux tile kernel: the parts of a tile behind / beside each other")
    ap.add_argument("--ftile", default="", help="y/z flux tile kernel: cells per y tile,levels per z tile (0 = automatic each)")
    ap.add_argument("--xexchange", default="auto", choices=("auto", "lds", "shuffle"),
                    help="x tile kernels: neighbouring cells exchange through LDS + barriers / by wavefront shuffles (a line inside one wavefront)")
    ap.add_argument("--launcher", default="python", choices=("python", "cpp"),
                    help="python: one process per GPU, dt exchange through torch.distributed (RCCL); cpp: examples/driver --gpus N, one "
                         "host thread per GPU in ONE process, dt exchange over N host doubles")
    ap.add_argument("--tuning", default="", help="want_units,two_phase_below,split_below (wavefront thresholds of the sweep launches)")
    ap.add_argument("--graph", default="auto", choices=("auto", "on", "off"), help="timeStep replayed from a captured HIP graph")
    ap.add_argument("--indep", type=int, default=-1, choices=(-1, 0, 1),
                    help="member ranges of the fused stage: 1 every range runs its whole stage on its own stream, 0 the polynomial kernels of "
                         "all ranges share one compute stream (round 2's schedule); -1 (default): the library's default (1)")
    ap.add_argument("--perens", type=int, default=0, choices=(0, 1), help="1: every member on its own vertical grid (per-member WENO tables)")
    ap.add_argument("--tailfusion", default="auto", choices=("auto", "on", "off"),
                    help="NT > 1, member lanes: tracer phase 2 + pressure pass + vapour fix-up as one launch (on) or three (off)")
    ap.add_argument("--fold", default="auto", choices=("auto", "on", "off"),
                    help="3-D member-lane stage: the z sweep stores the y+z part of the state's divergence (on) or its own differences (off)")
    ap.add_argument("--detail", default="bench_detail.json", help="file (beside bench.py) that receives the full measurement object")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-worker", default="", help=argparse.SUPPRESS)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true")
    return ap


def parse_args(argv=None):
    return build_parser().parse_args(argv)


def main():
    args = parse_args()
    if args.cpu_worker:
        c, n, t = args.cpu_worker.split(",")
        print(json.dumps(cpu_worker(c, int(n), int(t))))
        return
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.launcher == "cpp":
        if "WORLD_SIZE" in os.environ and int(os.environ.get("RANK", "0")) != 0:
            return                      # under torch.distributed.run: rank 0 alone drives every device through the C++ path
        launch_cpp(args)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args.gpus)
        return
    worker(args)


if __name__ == "__main__":
    main()
