"""The bench line against the contract of the task: the LAST stdout line is one strict-JSON object of less than 4 KB (the driver
parses that line; round 4's 32 KB object was recorded as `parsed: null`), the metric of BASELINE.json on the configuration it is
quoted on, `roofline` and `cpu_baseline` objects, self-consistent numbers; everything else is in the detail file beside bench.py.

GPU (-m gpu): `bench.py` is RUN in a child process and what it prints is validated --
  * once with the DRIVER's command shape (`--gpus 1 --steps 2 --warmup 1`, nothing disabled: CPU baseline, other configurations,
    module timings), which is the line that was too long in round 4;
  * the fast variants (one step, no CPU baseline, no other configs): default, small ensemble, C++ launcher.
Static (runs anywhere): the committed line + detail object of the round."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMITTED = "r06_c2_bench_default.json"
COMMITTED_DETAIL = "r06_c2_bench_detail.json"
LIMIT = 4096


def _strict(text):
    def bad(c):
        raise ValueError("non-strict JSON constant %r" % c)
    return json.loads(text, parse_constant=bad)


def _line(name):
    if not os.path.exists(os.path.join(ROOT, "profiles", name)):
        pytest.skip("profiles/%s: the round's default bench line has not been recorded yet" % name)
    lines = [l for l in open(os.path.join(ROOT, "profiles", name)).read().splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py prints ONE JSON line"
    assert name != COMMITTED or len(lines[0]) < LIMIT
    return _strict(lines[0])


def _check_contract(d, nens_per_gpu=1024, need_roofline=True):
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["unit"] == "cell-updates/s" and "cell-updates" in d["metric"] and "cell-updates" in json.dumps(base["metric"])
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"].startswith("synthetic")
    cfg = d["config"]
    assert list(cfg)[:2] == ["scaling", "nens_total"]                          # the N-GPU record is unambiguous from its first two keys
    assert "workload" in cfg and "model" not in cfg and cfg["nens_per_gpu"] == nens_per_gpu and (cfg["nx"], cfg["ny"], cfg["nz"]) == (32, 32, 60)
    # value = cells x sub-steps per timeStep / time per timeStep
    cells = cfg["nens_total"] * cfg["nx"] * cfg["ny"] * cfg["nz"]
    assert abs(d["value"] - cells * cfg["substeps_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-5 * d["value"]
    assert d["value"] > 0 and cfg["substeps_per_step"] >= 3
    if not need_roofline:
        return
    r = d["roofline"]
    assert r["bound"] in ("hbm", "fp64-valu") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    assert 0 < r["stage_frac"] <= r["frac"]          # the stage as a whole is never above its dominant kernel alone
    # achieved = algorithmic bytes of one launch / the kernel's measured duration; the bytes are SURVEY 8d's 384 B per cell-update
    # (NT = 1) x the cell-updates one launch (one stage of all cells = a third of a sub-step) covers
    assert abs(r["alg_bytes_per_launch"] - cells * 384 / 3.0) <= 1e-5 * r["alg_bytes_per_launch"]
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["ms_per_stage"] * 1e-3) / 1e9) <= 1e-4 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > r["alg_bytes_per_launch"]      # PMC bytes of the same kernel (null without a profile)
    assert 0 < r["valu"]["frac"] < 1


def _run(argv, detail):
    path = os.path.join(ROOT, detail)
    if os.path.exists(path):
        os.remove(path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--detail", detail] + list(argv), capture_output=True, text=True,
                       timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert lines, "bench.py printed nothing"
    last = lines[-1]
    assert len(last) < LIMIT, "the line the driver parses has %d bytes" % len(last)
    d = _strict(last)
    assert sum(1 for l in lines if l.startswith("{")) == 1, "bench.py prints ONE JSON line on stdout"
    full = _strict(open(path).read())
    os.remove(path)
    return d, full


def _run_bench(*extra):
    return _run(["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-other-configs"] + list(extra), "bench_detail_test.json")


@pytest.mark.gpu
def test_bench_py_the_drivers_command_prints_a_parseable_line():
    """`python3 bench.py --gpus 1 --steps K --warmup W` with NOTHING disabled: the command whose output the driver records"""
    d, full = _run(["--gpus", "1", "--steps", "2", "--warmup", "1"], "bench_detail_test.json")
    _check_contract(d)
    assert d["steps"] == 2 and d["warmup"] == 1
    c = d["cpu_baseline"]
    assert c is not None and c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == d["unit"] and c["value"] > 0 and c["sample"]
    assert c["one_core"]["value"] > 0 and c["processes_x_1_core"]["value"] >= c["one_core"]["value"]
    o = d["other"]
    for k in ("c3", "c4", "c2_perens", "c4_perens", "c4_full", "c2_shard128", "c2_limiter1", "c2_limiter2", "ref_nens1", "c2grid_nens1"):
        assert o[k] > 0, k
    assert o["c3_ms_per_step"] > 0 and o["c4_ms_per_step"] > 0
    # per-member vertical grids (the coupler's general contract) near the shared-table configurations: 0.88-0.91 x as
    # a rule; the guard leaves room below that: before the handles of a device shared their range streams the C4 shard's variant came out
    # at 0.75-0.80 x in ~1 of 5 bench runs (a degraded state of the process)
    assert o["c2_perens"] >= 0.80 * d["value"] and o["c4_perens"] >= 0.68 * o["c4"], (o["c2_perens"], d["value"], o["c4_perens"], o["c4"])
    for k in ("kessler_time_step", "sponge_layer", "compute_gcm_forcing_tendencies", "apply_gcm_forcing_tendencies"):
        assert o["modules_ms"][k] > 0
    for k in ("sponge_layer", "compute_gcm_forcing_tendencies", "apply_gcm_forcing_tendencies"):
        assert 0 < o["modules_hbm_frac"][k] < 1
    # the roofline that binds, as flat keys (null when no PMC profile of this build is committed)
    r = d["roofline"]
    for k in ("valu_frac_spec_clock", "valu_issue_frac", "sustained_clock_GHz", "stage_valu_frac"):
        assert k in r and (r[k] is None or r[k] > 0), k
    # the detail object holds what the line summarises
    assert abs(full["value"] - d["value"]) <= 1e-9 * d["value"]
    assert abs(full["other_configs"]["c3"]["value"] - o["c3"]) <= 1e-3 * o["c3"]
    assert full["roofline"]["frac_definition"] and full["kernel_rooflines"] and full["cpu_baseline"]["sample"]


@pytest.mark.gpu
def test_bench_py_runs_and_prints_a_valid_line():
    d, full = _run_bench()
    _check_contract(d)
    assert d["steps"] == 1 and d["warmup"] == 0 and d["cpu_baseline"] is None and "other" not in d
    names = [k["kernel"] for k in full["kernel_rooflines"]]
    assert "awfl_flux_kernel" in names and "awfl_xupd_kernel" in names          # the HIP kernels ran and were timed
    m = full["config"]["lane_mapping"]
    assert not m["yz_flat"] and not m["x_tiles"] and d["config"]["lanes"] == "member+xsweep"      # 1024 members: member lanes, sweeps


@pytest.mark.gpu
def test_bench_py_small_ensemble_line_uses_flat_lanes_and_tiles():
    d, full = _run_bench("--nens", "2")
    _check_contract(d, nens_per_gpu=2)
    m = full["config"]["lane_mapping"]
    assert m["yz_flat"] and m["x_tiles"] and m["flat_cells"] and d["config"]["lanes"] == "flat+xtile-shfl"      # (2 x 32 lanes: a line is one wavefront)


@pytest.mark.gpu
def test_bench_py_cpp_launcher_line():
    d, _ = _run_bench("--launcher", "cpp", "--nens", "32")
    _check_contract(d, nens_per_gpu=32, need_roofline=False)
    assert "examples/driver" in d["config"]["launcher"] and d["config"]["ranks_seen"] == 1 and d["roofline"] is None


def test_compact_line_of_a_large_detail_object_stays_below_the_limit():
    """bench.compact_line / emit on a synthetic worst case: long strings and NaNs in the detail object never reach the line"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    rf = {"bound": "fp64-valu", "kernel": "awfl_flux_kernel", "achieved": 1859.616683326781, "peak": 8000.0, "unit": "GB/s",
          "frac": 0.2324520854158476, "traffic": 16163791483.25926, "ms_per_stage": 4.33, "alg_bytes_per_launch": 8053063680.0,
          "stage_ms_back_to_back": 8.05, "stage_frac": 0.125, "stage_traffic": 3.76e10, "stage_traffic_ratio": 4.67,
          "frac_definition": "x" * 3000, "note": "y" * 3000, "valu": {"frac": 0.7, "issue_frac_at_sustained_clock": 0.9, "sustained_clock_GHz_approx": 1.85}}
    full = {"metric": "cell-updates/sec (AWFL dycore step)", "value": 2.6e9, "unit": "cell-updates/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "ms_per_step": 217.9, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"scaling": "weak", "nens_total": 1024, "nens_per_gpu": 1024, "workload": "w" * 120, "nx": 32, "ny": 32, "nz": 60,
                       "lane_mapping": {"yz_flat": False, "x_tiles": False, "tile": {"W": 64}}},
            "roofline": rf, "cpu_baseline": {"value": 3.5e6, "unit": "cell-updates/s", "cores": 16, "kind": "port", "cpu_model": "m" * 80,
                                             "one_core": {"value": 2.2e5, "seconds": 5.0, "nens": 2}, "sample": "s" * 2000},
            "kernels": {"k%d" % i: {"avg_ms": float("nan")} for i in range(50)}, "kernel_rooflines": [dict(rf) for _ in range(8)],
            "other_configs": dict([(k, {"value": 1e9, "roofline": dict(rf), "kernel_rooflines": [dict(rf)] * 6, "note": "n" * 500})
                                   for k in ("c3", "c4", "c4_full", "c2_shard128", "c2_limiter1", "c2_limiter2", "ref_nens1", "c2grid_nens1")]
                                  + [("modules", {"kessler_time_step": {"ms": 4.1, "bytes": 1.0, "note": "z" * 300}})])}
    line = json.dumps(b.compact_line(full), allow_nan=False, separators=(",", ":"))
    assert len(line) < LIMIT - 1024
    d = _strict(line)
    assert d["roofline"]["stage_traffic_ratio"] == 4.67 and d["other"]["c4"] == 1e9 and "frac_definition" not in d["roofline"]
    _strict(json.dumps(b._sig(full, 17), allow_nan=False))       # the detail object is strict JSON too (NaN -> null)


def test_committed_line_has_the_contract_fields():
    d = _line(COMMITTED)
    _check_contract(d)
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == d["unit"] and c["value"] > 0 and c["sample"]
    # one core, and P processes x one core
    assert c["one_core"]["value"] > 0 and c["processes_x_1_core"]["value"] >= c["one_core"]["value"]


def test_committed_line_other_configs():
    d = _line(COMMITTED)
    full = _line(COMMITTED_DETAIL)
    o = d["other"]
    for k in ("c3", "c4", "c4_full", "c2_shard128", "c2_limiter1", "c2_limiter2", "ref_nens1", "c2grid_nens1"):
        assert o[k] > 0, k
    # the limiter inputs flag rows and are slower than the smooth default
    oc = full["other_configs"]
    for k, lo in (("c2_limiter1", 0.1), ("c2_limiter2", 0.8)):
        assert oc[k]["fct_rows_flagged_last_stage"] >= lo * oc[k]["fct_rows"] and o[k] < d["value"]
    for k in ("kessler_time_step", "sponge_layer", "compute_gcm_forcing_tendencies", "apply_gcm_forcing_tendencies"):
        assert oc["modules"][k]["ms"] > 0 and oc["modules"][k]["bytes"] > 0
