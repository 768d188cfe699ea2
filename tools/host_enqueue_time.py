#!/usr/bin/env python3
"""How long the HOST needs to enqueue one timeStep against how long the GPU needs to run it (run on a GPU box):

  python tools/host_enqueue_time.py [--config c4] [--nens N] [--chunks K] [--steps S]

With dt_dyn_hint given, pam_amd_awfl_time_step contains no synchronisation: the call returns when every launch of the step has been
handed to the runtime.  `enqueue` = wall time of those calls with an idle GPU queue in front of them (the device is drained before each
call, so the runtime never blocks on a full queue); `gpu` = time per step of S steps issued back to back and synchronised once.
A stage of a small workload is a dozen launches of 10-100 us: where enqueue >= gpu, the host's launch rate is what bounds the step."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import bench
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c4")
    ap.add_argument("--nens", type=int, default=0)
    ap.add_argument("--chunks", type=int, default=-1)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    args = bench.parse_args(["--config", a.config, "--chunks", str(a.chunks)] + (["--nens", str(a.nens)] if a.nens else []))
    dev = torch.device("cuda", 0)
    job = bench.Job(a.config, args, dev, 0, 1, a.nens)
    d, c = job.dycore, job.coupler
    n = d.timeStep(c)                       # warm-up; also fixes dt_dyn
    dt = d.last_dt_dyn * 0.999
    for _ in range(2):
        d.timeStep(c, dt_dyn_hint=dt)
    torch.cuda.synchronize()
    enq = []
    for _ in range(a.steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d.timeStep(c, dt_dyn_hint=dt)
        enq.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        d.timeStep(c, dt_dyn_hint=dt)
    torch.cuda.synchronize()
    gpu = (time.perf_counter() - t0) / a.steps
    enq.sort()
    nsub = d.last_ncycles
    print("%s nens %d chunks %s: %d sub-steps; host enqueue %.3f ms per step (median; min %.3f), back to back %.3f ms per step -> %s"
          % (a.config, job.nens, a.chunks if a.chunks >= 0 else "auto", nsub, enq[len(enq) // 2] * 1e3, enq[0] * 1e3, gpu * 1e3,
             "HOST-bound" if enq[len(enq) // 2] > 0.9 * gpu else "device-bound"))


if __name__ == "__main__":
    main()
