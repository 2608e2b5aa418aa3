"""bench.py prints exactly one JSON line on stdout with the contract's keys (run small and fast)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--nx", "512", "--steps", "6", "--warmup", "2",
                        "--jacobi-sweeps-timed", "20", "--cpu-seconds", "0.5"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "cell-updates/s" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert abs(d["value"] - 512 * 512 * 6 / (d["ms_per_step"] * 6e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["algorithmic_bytes_per_launch"] == 3 * 8 * 512 * 512
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "oracle/vof_oracle.c" in cb["sample"]
    assert "workload" in d["config"] and "model" not in d["config"]


def test_bench_refuses_a_gpu_count_that_does_not_match_the_launcher():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode != 0 and "torch.distributed.run" in (r.stderr + r.stdout)
