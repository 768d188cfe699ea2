"""The coupler-module timings of bench.py alone (Kessler, sponge layer, GCM forcing at the C2 grid): one JSON object on stdout.
Run on the GPU box:  python tools/bench_modules.py"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    out = bench.modules_timing(torch, dev)
    print(json.dumps({k: ({kk: vv for kk, vv in v.items() if kk != "note"} if isinstance(v, dict) else v) for k, v in out.items()}))
