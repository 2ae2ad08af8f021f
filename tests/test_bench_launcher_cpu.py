"""CPU: `python bench.py --gpus N` with no rendezvous in the environment must start N rank processes itself (fresh children, gloo
barrier + max-over-ranks) and relay ONE JSON line with n_gpus == N.  The GPU chain is replaced by a host sleep (--stub-step-ms):
this checks the launcher and the multi-rank protocol of bench.main, not the kernels (BASELINE configs[4])."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env_extra=None, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_2_launches_two_ranks_and_reports_them():
    r = run_bench("--gpus", "2", "--steps", "5", "--warmup", "1", "--stub-step-ms", "20")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["stub"] is True and d["n_gpus"] == 2 and d["steps"] == 5 and d["scaling"] == "weak"
    # rank 1 sleeps 40 ms per step, rank 0 sleeps 20: the reported time is the max over ranks, the value counts both ranks' frames
    assert 38.0 <= d["ms_per_step"] <= 80.0, d
    assert abs(d["value"] - 2 * 5 / (d["ms_per_step"] * 5e-3)) < 1e-6 * d["value"]


def test_gpus_1_is_a_single_process():
    r = run_bench("--gpus", "1", "--steps", "3", "--warmup", "0", "--stub-step-ms", "5")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1


def test_world_size_mismatch_fails_loudly():
    # an external launcher with a different number of ranks than --gpus: refuse instead of silently measuring something else
    r = run_bench("--gpus", "4", "--steps", "1", "--stub-step-ms", "1", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_a_failing_rank_fails_the_run():
    # without a GPU the real chain refuses to run in every rank: the launcher must turn that into a non-zero exit, not a hang
    r = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", timeout=300)
    import torch
    if torch.cuda.is_available():
        assert r.returncode == 0
    else:
        assert r.returncode != 0 and "exited with code" in r.stderr
