"""bench.py's N > 1 code path ON a GPU box (the scaling bench itself is the driver's to run, on a whole node):

  * the driver's launch line for N = 2 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 2 --steps K --warmup W` -- on this box's ONE GPU: both ranks share the card, so bench.py takes its
    rehearsal branch for the 8-byte dt exchange (gloo: RCCL refuses two ranks on one device); everything else -- rank plumbing,
    per-rank member shards, the barrier + max-over-ranks timing, the aggregate value, ONE parseable line from rank 0 -- is the code
    the driver's 8-GPU run executes;
  * the RCCL calls themselves (process group with `device_id`, all-reduce(MIN) of dt on the device, barrier) with a single rank
    (PAM_AMD_DIST_SELFTEST=1): the only way to run them on a 1-GPU box.
Both use a small member count so that two ranks on one card finish in seconds."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
LIMIT = 4096


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(**kw)
    return env


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _last_json_line(r):
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line, the other ranks none: %r" % (r.stdout[-1500:],)
    assert len(lines[0]) < LIMIT
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_driver_launch_line_two_ranks(scaling, tmp_path):
    nens = 128
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--nens", str(nens),
           "--scaling", scaling, "--detail", str(tmp_path / "detail.json")]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    d = _last_json_line(r)
    cfg = d["config"]
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == scaling
    assert list(cfg)[:2] == ["scaling", "nens_total"] and cfg["ranks_seen"] == 2
    if scaling == "weak":          # per-GPU work fixed: every rank runs `nens` members
        assert cfg["nens_total"] == 2 * nens and cfg["nens_per_gpu"] == nens
    else:                          # total work fixed: the members are split
        assert cfg["nens_total"] == nens and cfg["nens_per_gpu"] == nens // 2
    assert "all-reduce(MIN) of dt" in cfg["collective"]
    # whole-job aggregate: the cells of ALL ranks x sub-steps / the slowest rank's time
    cells = cfg["nens_total"] * cfg["nx"] * cfg["ny"] * cfg["nz"]
    assert abs(d["value"] - cells * cfg["substeps_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-5 * d["value"]
    assert d["value"] > 0 and d["cpu_baseline"] is None           # (the CPU baseline is an N = 1 extra)
    # (the compact line rounds to a few significant digits)
    assert cfg["rank_ms_per_step"]["min"] <= cfg["rank_ms_per_step"]["max"] <= d["ms_per_step"] * (1 + 1e-4)


@pytest.mark.gpu
def test_rccl_path_with_one_rank(tmp_path):
    cmd = [sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--warmup", "1", "--nens", "128", "--no-cpu-baseline",
           "--no-other-configs", "--detail", str(tmp_path / "detail.json")]
    r = subprocess.run(cmd, env=_env(PAM_AMD_DIST_SELFTEST="1", MASTER_PORT=str(_port())), capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    d = _last_json_line(r)
    assert d["n_gpus"] == 1 and d["config"]["ranks_seen"] == 1 and d["value"] > 0
    # the same job without the process group: the exchange must not change what is computed (same sub-step count)
    r2 = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    d2 = _last_json_line(r2)
    assert d2["config"]["substeps_per_step"] == d["config"]["substeps_per_step"]
