"""Does the throughput of a two-range workload depend on how many streams the process created before the handle's own?  (Hypothesis for
the slow mode of c4_perens / c2_shard128 inside bench.py: the two ranges' streams landing on one hardware queue.)
    python tools/probe/stream_parity.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

import bench  # noqa: E402


def run(cfg, perens, steps, warmup, nens=0):
    a = bench.parse_args(["--gpus", "1", "--no-kernel-timing"])
    j = bench.Job(cfg, a, torch.device("cuda:0"), 0, 1, nens, perens=perens)
    u, el, _ = j.timed(steps, warmup)
    j.close()
    del j
    return u / el / 1e9


if __name__ == "__main__":
    keep = []
    for n in range(0, 10):
        v1 = run("c4", True, 20, 3)
        v2 = run("c2", False, 10, 2, nens=128)
        print("extra streams alive %d: c4_perens %.4f G   c2_shard128 %.4f G" % (len(keep), v1, v2), flush=True)
        keep.append(torch.cuda.Stream())
        with torch.cuda.stream(keep[-1]):
            torch.zeros(1, device="cuda:0").add_(1)
