"""The committed bench line of the round (profiles/r03_c2_bench_default.json: `python bench.py` on one MI355X) against the contract of
the task: one JSON object per line, the metric of BASELINE.json on the configuration it is quoted on, `roofline` and `cpu_baseline`
objects, self-consistent numbers.  Static data: runs anywhere."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    lines = [l for l in open(os.path.join(ROOT, "profiles", name)).read().splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py prints ONE JSON line"
    return json.loads(lines[0])


def test_default_line_has_the_contract_fields():
    d = _line("r03_c2_bench_default.json")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["unit"] == "cell-updates/s" and "cell-updates" in d["metric"] and "cell-updates" in json.dumps(base["metric"])
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"].startswith("synthetic")
    cfg = d["config"]
    assert "workload" in cfg and "model" not in cfg and cfg["nens_per_gpu"] == 1024 and (cfg["nx"], cfg["ny"], cfg["nz"]) == (32, 32, 60)
    # value = cells x sub-steps per timeStep / time per timeStep
    cells = cfg["nens_total"] * cfg["nx"] * cfg["ny"] * cfg["nz"]
    assert abs(d["value"] - cells * cfg["substeps_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "fp64-valu") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # achieved = algorithmic bytes of one launch / the kernel's measured duration; the bytes are SURVEY 8d's 384 B per cell-update
    # (NT = 1) x the cell-updates one launch (one stage of all cells = a third of a sub-step) covers
    assert abs(r["alg_bytes_per_launch"] - cells * 384 / 3.0) < 1.0
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["ms_per_stage"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > r["alg_bytes_per_launch"]      # PMC bytes of the same kernel (null without a profile)
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == d["unit"] and c["value"] > 0 and c["sample"]


def test_limiter_lines_flag_rows_and_are_slower():
    base = _line("r03_c2_bench_default.json")
    for name, lo in (("r03_c2_bench_limiter1.json", 0.1), ("r03_c2_bench_limiter2.json", 0.8)):
        d = _line(name)
        cfg = d["config"]
        assert cfg["limiter_input"] in (1, 2) and cfg["fct_rows_flagged_last_stage"] >= lo * cfg["fct_rows"]
        assert d["value"] < base["value"]
