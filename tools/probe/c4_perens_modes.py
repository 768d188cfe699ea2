"""Where does the slow mode of the per-member C4 shard come from?  Times a fresh Job of it repeatedly inside ONE process, optionally with
another workload's Job created, run and closed in front (as bench.py's `other` sequence does).  Run on the GPU box:
    python tools/probe/c4_perens_modes.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

import bench  # noqa: E402


def run(cfg, perens, steps, warmup, nens=0):
    a = bench.parse_args(["--gpus", "1", "--no-kernel-timing"])
    j = bench.Job(cfg, a, torch.device("cuda:0"), 0, 1, nens, perens=perens)
    u, el, _ = j.timed(steps, warmup)
    lm = j.dycore.get_lane_mapping() if hasattr(j.dycore, "get_lane_mapping") else None
    j.close()
    del j
    return u / el / 1e9, lm


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    for front in (None, ("c4", False), ("c2", True), ("c2", True), None, ("c2", False)):
        if front:
            v, _ = run(front[0], front[1], 3, 1)
            print("front %-3s perens=%d  %.4f G" % (front[0], front[1], v), flush=True)
        for _ in range(2):
            v, lm = run("c4", True, 30, 3)
            print("   c4 perens %.4f G  %s" % (v, lm), flush=True)
