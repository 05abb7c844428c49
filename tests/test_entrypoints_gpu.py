"""GPU: the reference-shaped entry points end to end on tiny recipes: `run.py` (compose -> generator -> fused training steps ->
model.safetensors + last.ckpt), resume from the periodic checkpoint, `run_inference.py` (pruned checkpoint -> hipGraph forward ->
uint8 predictions), and the UNETR recipe with its shipped dropout."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, ok=True):
    p = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=900)
    if ok:
        assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    return p


def test_run_train_resume_and_inference_tiny_recipe():
    logs = os.path.join(ROOT, "logs")
    for f in ("model.safetensors", "last.ckpt"):
        if os.path.exists(os.path.join(logs, f)):
            os.remove(os.path.join(logs, f))
    common = ["+default_configs=tiny", "++data.tile_size=128", "++train.batch_size=2"]
    p = _run(["run.py"] + common + ["++train.max_steps=6", "++train.checkpoint_every=3"])
    assert "step     0" in p.stdout and "tiles/s" in p.stdout
    ck = torch.load(os.path.join(logs, "last.ckpt"), map_location="cpu", weights_only=True)
    assert ck["global_step"] == 6 and ck["optimizer_state"]["step"] == 6 and ck["bn_running_stats"] == "rank-local"
    assert any(k.startswith("generator.decoder.") for k in ck["state_dict"]) and "loss_reconstruct.marker_weights" in ck["state_dict"]
    from safetensors.torch import load_file
    sd = load_file(os.path.join(logs, "model.safetensors"))
    assert sd and all((".lora" in k) or not k.startswith("encoder.vit.") for k in sd)
    # resume: continues at step 6 with the saved Adam state
    p = _run(["run.py"] + common + ["++train.max_steps=8", "++train.resume_from=" + os.path.join(logs, "last.ckpt")])
    assert "resumed from" in p.stdout and "at step 6" in p.stdout and "2 steps" in p.stdout
    # inference: refuses a random frozen encoder unless told so, then writes uint8 predictions
    bad = _run(["run_inference.py", "--checkpoint_dir", logs, "--batch_size", "2", "--num_batches", "1"] + common, ok=False)
    assert bad.returncode != 0 and "encoder_weights is null" in (bad.stderr + bad.stdout)
    out_dir = os.path.join(logs, "predictions_test")
    _run(["run_inference.py", "--checkpoint_dir", logs, "--batch_size", "2", "--num_batches", "1", "--output_dir", out_dir,
          "--allow-random-encoder"] + common)
    pred = np.load(os.path.join(out_dir, "batch_0000.npy"))
    assert pred.dtype == np.uint8 and pred.shape == (2, 3, 128, 128) and pred.std() > 0


def test_run_unetr_recipe_with_shipped_dropout():
    p = _run(["run.py", "+default_configs=unetr", "++model.encoder.encoder_name=tiny4_swiglu", "++data.tile_size=128",
              "++data.targ_channel_names=[Hoechst,CD31,CD45]", "++train.batch_size=2", "++train.max_steps=4"])
    assert "step     3" in p.stdout
    losses = [float(l.split("loss")[1].split()[0]) for l in p.stdout.splitlines() if l.startswith("step")]
    assert all(np.isfinite(losses))


def test_run_py_two_ranks_rehearsal_on_one_gpu():
    """The multi-rank branch of run.py (rendezvous, parameter broadcast, bucketed exchange inside training_step, rank-0 checkpoint
    writes): two ranks share cuda:0 and exchange over gloo -- RCCL refuses two ranks on one device, the transport is the only
    difference to the 8-GPU command line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MIPHEI_DIST_BACKEND="gloo", MIPHEI_SHARE_GPU="1", OMP_NUM_THREADS="2")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), "run.py", "+default_configs=tiny", "++data.tile_size=128",
                        "++train.batch_size=2", "++train.max_steps=4", "++train.checkpoint_every=2"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert p.stdout.count("step     0") == 1 and "tiles/s" in p.stdout          # rank 0 alone reports
    ck = torch.load(os.path.join(ROOT, "logs", "last.ckpt"), map_location="cpu", weights_only=True)
    assert ck["global_step"] == 4
