"""Training entry point with the reference's command-line shape:

    python run.py +default_configs=miphei-vit ++train.max_steps=50
    python -m torch.distributed.run --nproc-per-node 8 run.py +default_configs=miphei-vit ++train.devices=8

Mirrors ``/root/reference/run.py`` + ``src/train.py:34-210`` for the parts on the MI355X hot path: compose the config,
build generator / loss / ``ModelModule``, run the (fused) training steps, save ``model.safetensors`` (LoRA + decoder) with
the reference key names.  Data loading is outside the path: tiles are synthetic and generated on the device.
"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main(argv):
    from miphei_vit_amd.synthetic import synthetic_batch, synthetic_init_
    from miphei_vit_amd.checkpoint import save_checkpoint_atomic, save_pruned_safetensors
    from miphei_vit_amd.config import compose
    from miphei_vit_amd.generators import get_generator
    from miphei_vit_amd.loss import WeightedMSELoss, marker_weights_from_file
    from miphei_vit_amd.models import ModelModule
    from miphei_vit_amd.trainer import DataParallelSync

    cfg = compose(os.path.join(ROOT, "configs"), argv)
    # Config branches of the reference's train.py that switch the objective (src/train.py:118-150) and are NOT on this path
    # fail loudly instead of silently training with WeightedMSELoss
    losses = cfg.train.losses
    if losses.get("use_weighted_mae"):
        raise NotImplementedError("train.losses.use_weighted_mae: the foreground-weighted focal loss (reference src/train.py:"
                                  "118-132) is outside the MI355X hot path; WeightedMSELoss (use_weighted_mae: false) is")
    if (losses.get("cell_loss") or {}).get("use_loss"):
        raise NotImplementedError("train.losses.cell_loss.use_loss: the cell-level loss (reference src/train.py:144-150) is "
                                  "outside the MI355X hot path")
    # (train.use_cell_metrics -- on in the reference's shipped recipes through train=cell -- only adds validation-time CellMetrics
    # (src/train.py:111-114, models.py:233-241); run.py runs training steps only, the extractor lives in miphei_vit_amd.cells)
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    # MIPHEI_DIST_BACKEND=gloo MIPHEI_SHARE_GPU=1: rehearsal of the multi-rank branch on a one-GPU box (RCCL refuses two ranks
    # on one device; gloo takes device tensors): every rank uses cuda:0.  The product transport is RCCL ("nccl").
    backend = os.environ.get("MIPHEI_DIST_BACKEND", "nccl")
    if os.environ.get("MIPHEI_SHARE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    nc = len(cfg.data.targ_channel_names)
    S, B = int(cfg.data.tile_size), int(cfg.train.batch_size)
    with torch.device(dev):
        generator = get_generator(cfg.model.model_name, S, 3, nc, cfg)
    if not cfg.model.encoder.get("pretrained", True) and cfg.model.encoder.encoder_weights is None:
        synthetic_init_(generator, seed=0)
    stats_path = cfg.data.channel_stats_path or "configs/channel_stats_orion.json"
    if not os.path.isabs(stats_path):
        stats_path = os.path.join(ROOT, stats_path)
    weights = marker_weights_from_file(stats_path, cfg.data.targ_channel_names)   # train.py:137-142
    loss = WeightedMSELoss(cfg.train.losses.lambda_factor, weights)
    module = ModelModule(generator, None, cfg.train.learning_rate_g * B ** 0.5, cfg.train.learning_rate_d, loss,
                         gan_train=cfg.train.gan_train).to(dev)
    steps = int(cfg.train.max_steps)
    module.total_iters = steps
    if world > 1:
        sync = DataParallelSync(generator._engine)
        sync.broadcast_parameters(0)
        module.grad_sync = sync
    logdir = os.path.join(ROOT, "logs")
    os.makedirs(logdir, exist_ok=True)
    start = 0
    resume = cfg.train.get("resume_from")
    if resume:      # ++train.resume_from=logs/last.ckpt : weights, Adam moments + step count, LR-schedule position
        # plain tensors / dicts / scalars only: safe to load with weights_only=True
        module.load_checkpoint_state(torch.load(resume, map_location="cpu", weights_only=True))
        start = module.global_step_
        if rank == 0:
            print(f"resumed from {resume} at step {start} (LR horizon {module.total_iters} steps)", flush=True)
    every = int(cfg.train.get("checkpoint_every") or 0)
    t0 = time.perf_counter()
    for i in range(start, steps):
        x, y = synthetic_batch(1234 + rank * 1000 + i, B, S, nc, dev)
        out = module.training_step({"image": x, "target": y}, i)
        if rank == 0 and (i % 10 == 0 or i == steps - 1):
            print(f"step {i:5d}  loss {float(out):.4f}  lr {module.current_lr(i):.3e}", flush=True)
        if rank == 0 and every and (i + 1) % every == 0 and i + 1 < steps:
            # rank 0's replica: parameters are identical on all ranks, BatchNorm running statistics are rank-local (noted in
            # the file as "bn_running_stats": "rank-local")
            save_checkpoint_atomic(module.checkpoint_state(), os.path.join(logdir, "last.ckpt"))
    module.on_train_end()           # drains the asynchronous NaN guard
    torch.cuda.synchronize()
    if rank == 0:
        dt = time.perf_counter() - t0
        print(f"{steps - start} steps, {world * B * (steps - start) / dt:.1f} tiles/s")
        save_checkpoint_atomic(module.checkpoint_state(), os.path.join(logdir, "last.ckpt"))
        save_pruned_safetensors(generator, os.path.join(logdir, "model.safetensors"))
        print("saved", os.path.join(logdir, "model.safetensors"))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
