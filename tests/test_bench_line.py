"""The bench line against the contract of the task: one JSON object per line, the metric of BASELINE.json on the configuration it is
quoted on, `roofline` and `cpu_baseline` objects, self-consistent numbers.

GPU (-m gpu): `bench.py` is RUN in a child process (one step, no CPU baseline, no other configs) and the line it prints is
validated -- a regression of the bench (a missing field, an inconsistent number, a crash) fails here.  The C++ launcher's line
(`--launcher cpp`) is validated the same way.  Static (runs anywhere): the committed line of the round, which also carries
`cpu_baseline` and `other_configs`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMITTED = "r04_c2_bench_default.json"


def _line(name):
    if not os.path.exists(os.path.join(ROOT, "profiles", name)):
        pytest.skip("profiles/%s: the round's default bench line has not been recorded yet" % name)
    lines = [l for l in open(os.path.join(ROOT, "profiles", name)).read().splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py prints ONE JSON line"
    return json.loads(lines[0])


def _check_contract(d, nens_per_gpu=1024, need_roofline=True):
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["unit"] == "cell-updates/s" and "cell-updates" in d["metric"] and "cell-updates" in json.dumps(base["metric"])
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"].startswith("synthetic")
    cfg = d["config"]
    assert "workload" in cfg and "model" not in cfg and cfg["nens_per_gpu"] == nens_per_gpu and (cfg["nx"], cfg["ny"], cfg["nz"]) == (32, 32, 60)
    # value = cells x sub-steps per timeStep / time per timeStep
    cells = cfg["nens_total"] * cfg["nx"] * cfg["ny"] * cfg["nz"]
    assert abs(d["value"] - cells * cfg["substeps_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert d["value"] > 0 and cfg["substeps_per_step"] >= 3
    if not need_roofline:
        return
    r = d["roofline"]
    assert r["bound"] in ("hbm", "fp64-valu") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert "frac_definition" in r and 0 < r["stage_frac"] <= r["frac"]       # the stage as a whole is never above its dominant kernel alone
    # achieved = algorithmic bytes of one launch / the kernel's measured duration; the bytes are SURVEY 8d's 384 B per cell-update
    # (NT = 1) x the cell-updates one launch (one stage of all cells = a third of a sub-step) covers
    assert abs(r["alg_bytes_per_launch"] - cells * 384 / 3.0) < 1.0
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["ms_per_stage"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > r["alg_bytes_per_launch"]      # PMC bytes of the same kernel (null without a profile)


def _run_bench(*extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                        "--no-other-configs"] + list(extra), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "bench.py prints ONE JSON line"
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_py_runs_and_prints_a_valid_line():
    d = _run_bench()
    _check_contract(d)
    assert d["steps"] == 1 and d["warmup"] == 0 and d["cpu_baseline"] is None and d["other_configs"] is None
    names = [k["kernel"] for k in d["kernel_rooflines"]]
    assert "awfl_flux_kernel" in names and "awfl_xupd_kernel" in names          # the HIP kernels ran and were timed
    m = d["config"]["lane_mapping"]
    assert not m["yz_flat"] and not m["x_tiles"]                                # 1024 members: member lanes, sweeps


@pytest.mark.gpu
def test_bench_py_small_ensemble_line_uses_flat_lanes_and_tiles():
    d = _run_bench("--nens", "2")
    _check_contract(d, nens_per_gpu=2)
    m = d["config"]["lane_mapping"]
    assert m["yz_flat"] and m["x_tiles"] and m["flat_cells"]


@pytest.mark.gpu
def test_bench_py_cpp_launcher_line():
    d = _run_bench("--launcher", "cpp", "--nens", "32")
    _check_contract(d, nens_per_gpu=32, need_roofline=False)
    assert "examples/driver" in d["config"]["launcher"] and d["config"]["ranks_seen"] == 1 and d["roofline"] is None


def test_committed_line_has_the_contract_fields():
    d = _line(COMMITTED)
    _check_contract(d)
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == d["unit"] and c["value"] > 0 and c["sample"]
    # one core, and P processes x one core
    assert c["one_core"]["cores"] == 1 and c["one_core"]["value"] > 0
    assert c["processes_x_1_core"]["processes"] == c["cores"] and c["processes_x_1_core"]["value"] >= c["one_core"]["value"]


def test_committed_line_other_configs():
    d = _line(COMMITTED)
    o = d["other_configs"]
    for k in ("c3", "c4", "c4_full", "c2_shard128", "c2_limiter1", "c2_limiter2", "ref_nens1", "c2grid_nens1"):
        assert o[k]["value"] > 0, k
    # the limiter inputs flag rows and are slower than the smooth default
    for k, lo in (("c2_limiter1", 0.1), ("c2_limiter2", 0.8)):
        assert o[k]["fct_rows_flagged_last_stage"] >= lo * o[k]["fct_rows"] and o[k]["value"] < d["value"]
    for k in ("kessler_time_step", "sponge_layer", "compute_gcm_forcing_tendencies", "apply_gcm_forcing_tendencies"):
        assert o["modules"][k]["ms"] > 0 and o["modules"][k]["bytes"] > 0
