"""bench.py's driver contract on the GPU box: ONE line on stdout, valid JSON with the fields the round driver reads
(metric/value/unit/n_gpus/steps/warmup/ms_per_step/..., roofline and cpu_baseline objects), also when RCCL is initialised
(its C-stdio version banner used to land on stdout after the JSON line)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(extra, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1"] + extra, cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_single_json_line_with_roofline_and_cpu_baseline():
    r = _run([])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 1 and r["unit"] == "tiles/s" and r["value"] > 0
    assert r["vs_baseline"] is None and r["scaling"] == "weak" and r["data"] == "synthetic" and "workload" in r["config"]
    rf = r["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert r["parity"]["worst_channel_rel_mse"] < r["parity"]["tolerance_rel_mse"]


def test_bench_stdout_stays_one_line_with_rccl():
    r = _run(["--no-cpu-baseline"], {"MIPHEI_FORCE_DDP": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29547"})
    assert r["value"] > 0 and r["config"]["parallelism"] == "dp1"
