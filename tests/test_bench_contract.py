"""bench.py's output contract (one JSON line with the driver's keys + roofline + cpu_baseline) and its multi-rank
path, rehearsed with two gloo ranks sharing the one GPU of the box."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gpu = pytest.mark.gpu

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _last_json(out: str):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


@gpu
def test_single_gpu_line():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "48", "--warmup", "8", "--envs", "512"],
                         capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert REQUIRED <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 48 and d["warmup"] == 8 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["value"] > 1e6 and abs(d["value"] - 512 * 48 / (d["ms_per_step"] * 48e-3)) / d["value"] < 0.02
    assert "workload" in d["config"] and d["vs_baseline"] is None
    r, c = d["roofline"], d["cpu_baseline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["parity_with_gpu"] is True and c["sample"]
    rep = d["ms_per_step_repeats"]
    assert rep["n"] == 5 and rep["min"] <= rep["median"] <= rep["max"] and rep["min"] <= d["ms_per_step"] <= rep["max"]
    assert d["world_size_observed"] == 1 and d["per_rank_env_steps_per_sec"] == [d["value"]]


@gpu
def test_two_rank_path_with_gloo():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "32", "--warmup", "4",
           "--envs", "512", "--backend", "gloo", "--device-index", "0", "--adv-allgather"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["cpu_baseline"] is None
    assert abs(d["value"] - 2 * 512 * 32 / (d["ms_per_step"] * 32e-3)) / d["value"] < 0.02  # whole-job aggregate


@gpu
def test_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it (the form the driver uses): the parent starts two ranks,
    rank 0 prints the one line for the whole job.  gloo because the box has one GPU (both ranks share it)."""
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "32", "--warmup", "4",
                          "--envs", "512", "--backend", "gloo", "--repeats", "2"],
                         capture_output=True, text=True, timeout=600, cwd=REPO,
                         env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # one line for the job, not one per rank
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size_observed"] == 2 and d["backend"] == "gloo"
    assert len(d["per_rank_env_steps_per_sec"]) == 2 and d["adv_allgather_bytes_per_rank"] == 16 * 512 * 4
    assert abs(d["value"] - 2 * 512 * 32 / (d["ms_per_step"] * 32e-3)) / d["value"] < 0.02
    assert d["value"] <= sum(d["per_rank_env_steps_per_sec"]) * 1.0001  # max over ranks bounds the aggregate


def test_gpus_flag_spawns_ranks_and_mismatch_fails_loudly():
    """No GPU needed: without a GPU every started rank must refuse to run (no CPU fallback), and a launcher whose
    world size disagrees with --gpus is an error rather than a silently smaller job."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                         timeout=300, cwd=REPO, env=dict(env, WORLD_SIZE="2", RANK="0"))
    assert out.returncode != 0 and "--gpus 4" in out.stderr and "WORLD_SIZE=2" in out.stderr
    import torch
    if torch.cuda.is_available():
        return
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo"],
                         capture_output=True, text=True, timeout=300, cwd=REPO, env=env)
    assert out.returncode != 0
    assert out.stderr.count("bench.py needs an MI355X") >= 2, out.stderr[-2000:]  # both ranks were started
