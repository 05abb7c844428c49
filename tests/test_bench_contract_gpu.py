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
    assert rf["traffic"] is None or "static" in rf["traffic_source"]
    rs = r["roofline_step"]
    assert rs["bound"] == "mfma" and abs(rs["frac"] - r["model_flops_frac"]) < 1e-3 and 0 < rs["frac"] < 1
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert "median of 5" in cb["sample"] and cb["cpu_model"] and cb["host_cores"] >= cb["cores"]
    assert cb["tiny"]["value"] > 0 and "configs[0]" in cb["tiny"]["sample"]
    assert r["parity"]["worst_channel_rel_mse"] < r["parity"]["tolerance_rel_mse"]
    assert r["parity"]["batch"] == 16           # parity leg runs at the timed batch (the 256-row tile path)
    assert r["value"] / cb["value"] >= 10       # north-star: >= 10x the reference CPU path on one MI355X


def test_bench_stdout_stays_one_line_with_rccl():
    r = _run(["--no-cpu-baseline"], {"MIPHEI_FORCE_DDP": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29547"})
    assert r["value"] > 0 and r["config"]["parallelism"] == "dp1"
    # the bucketed exchange ran on RCCL (one rank): rank count and the exposed-communication figure are on the line
    assert r["rccl_ranks"] == 1 and r["config"]["lora_buckets"] == 4 and r["exposed_comm_ms_per_step"] >= 0


def test_bench_inference_b64_hipgraph_line_has_p50():
    """BASELINE configs[4]: hipGraph-captured forward at batch 64, tiles/s + p50 batch latency on the driver-readable line."""
    r = _run(["--mode", "infer", "--batch", "64"])
    assert r["config"]["hipgraph"] is True and r["config"]["global_batch"] == 64 and "configs[4]" in r["config"]["workload"]
    assert r["p50_batch_latency_ms"] > 0 and r["value"] > 0
    assert abs(r["p50_batch_latency_ms"] - r["ms_per_step"]) < 0.25 * r["ms_per_step"]   # the device is never idle
    assert r["roofline_step"]["frac"] > 0.2


def test_bench_512_line_has_roofline():
    """BASELINE configs[3] (1-GPU leg): 512x512 tiles, 1301 tokens."""
    r = _run(["--img", "512", "--batch", "4", "--no-cpu-baseline"])
    assert r["config"]["img"] == 512 and "configs[3]" in r["config"]["workload"] and r["value"] > 0
    assert r["roofline"]["bound"] == "mfma" and r["roofline"]["launches"] > 0 and 0 < r["roofline"]["frac"] < 1
    assert r["roofline_step"]["flops_per_tile"] == 7604.0e9
