"""bench.py's rank plumbing, on CPU (no GPU, no dycore: PAM_AMD_BENCH_DRYRUN=1 stops after the rendezvous).
`python bench.py --gpus N` without WORLD_SIZE must start N ranks itself (fresh children; the parent never touches a GPU);
under a launcher (WORLD_SIZE set) it must refuse a --gpus that disagrees with the world size instead of silently
measuring something else (round-1 finding: the flag was parsed and ignored)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(PAM_AMD_BENCH_DRYRUN="1", **kw)
    return env


def test_gpus_flag_spawns_that_many_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--scaling", "strong"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2.0 and line["scaling"] == "strong"


def test_eight_ranks_strong_scaling_of_c2_is_128_members_each():
    """the driver's N = 8 line: `bench.py --gpus 8 --scaling strong --config c2` starts 8 ranks, every one of them is seen by the
    collective, and each owns 128 of the 1024 members (BASELINE config C2 sharded by nens)"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--scaling", "strong", "--config", "c2"], env=_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8.0
    assert line["shard_sizes"] == [128] * 8 and line["nens_total"] == 1024


def test_eight_ranks_weak_scaling_keeps_the_config_per_rank():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--config", "c4"], env=_env(), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["ranks_seen"] == 8.0 and line["shard_sizes"] == [512] * 8      # C4: 512 = one GPU's shard of nens = 4096


def test_single_rank_default():
    r = subprocess.run([sys.executable, BENCH], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_world_size_mismatch_is_an_error():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8"], env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_parent_of_a_multi_rank_run_never_imports_torch():
    src = open(BENCH).read()
    head = src[:src.index("def worker(")]
    assert "import torch" not in head.replace("        import torch", "")   # only Job/worker (children) import it
