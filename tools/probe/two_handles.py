"""Does a second, idle handle in the process slow a workload down?  (bench.py kept its main Job alive while it timed the other
configurations for one experiment of round 6: every one of them lost 10-35 %.)   python tools/probe/two_handles.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

import bench  # noqa: E402


def job(cfg, perens=False, nens=0, chunks=-1):
    a = bench.parse_args(["--gpus", "1", "--no-kernel-timing", "--chunks", str(chunks)])
    return bench.Job(cfg, a, torch.device("cuda:0"), 0, 1, nens, perens=perens)


def timed(cfg, perens=False, nens=0, steps=30, warmup=3):
    j = job(cfg, perens, nens)
    u, el, _ = j.timed(steps, warmup)
    j.close()
    del j
    return u / el / 1e9


if __name__ == "__main__":
    print("alone:                         c4 %.4f  c4_perens %.4f  c2@128 %.4f" % (timed("c4"), timed("c4", True), timed("c2", nens=128, steps=10)), flush=True)
    for chunks, label in ((-1, "automatic ranges"), (1, "one range")):
        idle = job("c2", nens=128, chunks=chunks)
        idle.timed(2, 1)
        print("idle C2@128 handle alive (%s): c4 %.4f  c4_perens %.4f  c2@128 %.4f" % (label, timed("c4"), timed("c4", True), timed("c2", nens=128, steps=10)), flush=True)
        idle.close()
        del idle
        print("after closing it:              c4 %.4f  c4_perens %.4f  c2@128 %.4f" % (timed("c4"), timed("c4", True), timed("c2", nens=128, steps=10)), flush=True)
    keep = [torch.empty(int(30e9 // 8), dtype=torch.float64, device="cuda:0")]
    print("30 GB of other data resident:  c4 %.4f  c4_perens %.4f  c2@128 %.4f" % (timed("c4"), timed("c4", True), timed("c2", nens=128, steps=10)), flush=True)
