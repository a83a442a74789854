"""bench.py's output contract (one JSON line with the driver's keys + roofline + cpu_baseline) and its multi-rank
path, rehearsed with two gloo ranks sharing the one GPU of the box."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _last_json(out: str):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


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


def test_two_rank_path_with_gloo():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "32", "--warmup", "4",
           "--envs", "512", "--backend", "gloo", "--device-index", "0", "--adv-allgather"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["cpu_baseline"] is None
    assert abs(d["value"] - 2 * 512 * 32 / (d["ms_per_step"] * 32e-3)) / d["value"] < 0.02  # whole-job aggregate
