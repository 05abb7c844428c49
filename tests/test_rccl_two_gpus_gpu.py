"""GPU, boxes with >= 2 devices only: the first real RCCL run with more than one rank -- `bench.py --gpus 2` launches its own ranks
(one per GPU, `torch.distributed` backend "nccl" = RCCL over xGMI), runs the bucketed gradient exchange from inside the backward
pass and reports it on the line; a second script checks that both ranks end a training run with identical parameters.  Skipped
on the one-GPU build boxes (there the exchange logic is covered by tests/test_ddp_two_ranks_gpu.py over gloo on a shared device
and by the gloo world-2 CPU tests).  What DP replaces: the reference trains with `devices=1` (/root/reference/src/train.py:205-207)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _need_two():
    if torch.cuda.device_count() < 2:          # (device_count does not initialise the GPU on this image)
        pytest.skip("needs >= 2 GPUs: real RCCL refuses two ranks on one device")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_over_rccl():
    _need_two()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--master-port", str(_free_port())], cwd=ROOT, capture_output=True, text=True, timeout=1200, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-1500:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["rccl_ranks"] == 2 and r["scaling"] == "weak" and r["config"]["global_batch"] == 32
    assert r["exposed_comm_ms_per_step"] is not None and r["exposed_comm_ms_per_step"] < 1.0, r["exposed_comm_ms_per_step"]
    assert len(r["comm_buckets"]) == 5 and r["value"] > 0


_RANK_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["MVIT_ROOT"])
rank, local = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", device_id=dev)
from miphei_vit_amd.generators import get_vitmatte
from miphei_vit_amd.loss import WeightedMSELoss
from miphei_vit_amd.models import ModelModule
from miphei_vit_amd.synthetic import synthetic_batch, synthetic_init_
from miphei_vit_amd.trainer import DataParallelSync
with torch.device(dev):
    model = get_vitmatte("tiny_swiglu", 128, 3, use_lora=True, pretrained=False)
torch.cuda.manual_seed(100 + rank)          # different initial weights per rank: the broadcast has to make them equal
synthetic_init_(model, seed=100 + rank)
mod = ModelModule(model, None, 2e-4, 0., WeightedMSELoss(50.0, torch.ones(3))).to(dev)
mod.total_iters = 100
mod.update_pix_metrics = False
sync = DataParallelSync(model._engine)
sync.broadcast_parameters(0)
mod.grad_sync = sync
for i in range(3):
    x, y = synthetic_batch(7 + 1000 * rank + i, 2, 128, 3, dev)      # a different minibatch per rank
    mod.training_step({"image": x, "target": y}, i)
mod.on_train_end()
flat = model._engine._ensure_flat().flat
mine = flat.detach().clone()
lo, hi = mine.clone(), mine.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN)
dist.all_reduce(hi, op=dist.ReduceOp.MAX)
ok = bool(torch.equal(lo, hi)) and bool(torch.isfinite(mine).all())
print(f"RANK{rank} identical={ok} n={mine.numel()}", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 3)
"""


def test_two_ranks_hold_identical_parameters_after_training(tmp_path):
    _need_two()
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MVIT_ROOT=ROOT, OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), str(script)], cwd=ROOT, capture_output=True, text=True,
                       timeout=900, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert "RANK0 identical=True" in p.stdout and "RANK1 identical=True" in p.stdout
