"""Training entry point with the reference's command-line shape:

    python run.py +default_configs=miphei-vit ++train.max_steps=50
    python -m torch.distributed.run --nproc-per-node 8 run.py +default_configs=miphei-vit ++train.devices=8

Mirrors ``/root/reference/run.py`` + ``src/train.py:34-210`` for the parts on the MI355X hot path: compose the config,
build generator / loss / ``ModelModule``, run the (fused) training steps, save ``model.safetensors`` (LoRA + decoder) with
the reference key names.  Data loading is outside the path: tiles are synthetic and generated on the device, or -- with
``++data.uint8_tiles=<tiles.npz>`` (arrays ``image`` [N,H,W,3] and ``target`` [N,H,W,C], uint8) -- a tile set resident in HBM
that goes through the on-device input stage every step (``io_stage.TrainAugmenter``: RandomCrop / flips / CoarseDropout +
both normalisations, the reference's ``dataset.py:244-311, 458-483`` without the CPU loader).  ``++data.val_uint8_tiles=<val.npz>``
(``image``, ``target`` and, for ``train.use_cell_metrics``, ``nuclei`` [N,H,W] int32 + ``slide_name`` [N]) runs
``ModelModule.validation_step`` over that set after training, as ``src/train.py:111-114`` + ``models.py:233-241, 290-291`` do.
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
    from miphei_vit_amd.config import check_precision, compose
    from miphei_vit_amd.generators import get_generator
    from miphei_vit_amd.loss import WeightedMSELoss, marker_weights_from_file
    from miphei_vit_amd.models import ModelModule
    from miphei_vit_amd.trainer import DataParallelSync

    cfg = compose(os.path.join(ROOT, "configs"), argv)
    check_precision(cfg.train.get("precision", "bf16-mixed"))
    # Config branches of the reference's train.py that switch the objective (src/train.py:118-150) and are NOT on this path
    # fail loudly instead of silently training with WeightedMSELoss
    losses = cfg.train.losses
    if losses.get("use_weighted_mae"):
        raise NotImplementedError("train.losses.use_weighted_mae: the foreground-weighted focal loss (reference src/train.py:"
                                  "118-132) is outside the MI355X hot path; WeightedMSELoss (use_weighted_mae: false) is")
    if (losses.get("cell_loss") or {}).get("use_loss"):
        raise NotImplementedError("train.losses.cell_loss.use_loss: the cell-level loss (reference src/train.py:144-150) is "
                                  "outside the MI355X hot path")
    # (train.use_cell_metrics -- on in the reference's shipped recipes through train=cell -- adds validation-time CellMetrics
    # (src/train.py:111-114, models.py:233-241): built below when a validation tile set with nuclei masks is given)
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
    # resident uint8 tile sets (the arrays are small enough for HBM by construction: 288 GB hold ~230 k ORION tiles)
    import numpy as np
    tiles = val_tiles = None
    if cfg.data.get("uint8_tiles"):
        z = np.load(cfg.data.uint8_tiles)
        tiles = (torch.from_numpy(z["image"]).to(dev), torch.from_numpy(z["target"]).to(dev))
        if tiles[0].dtype != torch.uint8 or tiles[0].shape[3] != 3 or tiles[1].shape[3] != nc or min(tiles[0].shape[1:3]) < S:
            raise ValueError(f"data.uint8_tiles: need uint8 image [N,H,W,3] and target [N,H,W,{nc}] with H, W >= {S}")
    cell_metrics = None
    if cfg.data.get("val_uint8_tiles"):
        z = np.load(cfg.data.val_uint8_tiles)
        val_tiles = {k: z[k] for k in z.files}
        # every input check of the validation pass happens HERE, before the first training step: a malformed validation set must
        # not be discovered after the run (the weights would be lost with it)
        vi_, vt_ = val_tiles.get("image"), val_tiles.get("target")
        if vi_ is None or vt_ is None or vi_.dtype != np.uint8 or vi_.ndim != 4 or vi_.shape[3] != 3 or vt_.ndim != 4 \
                or vt_.shape[3] != nc or vt_.shape[:3] != vi_.shape[:3]:
            raise ValueError(f"data.val_uint8_tiles: need uint8 image [N,H,W,3] and target [N,H,W,{nc}] of one spatial size")
        if vi_.shape[1] != S or vi_.shape[2] != S:
            raise ValueError(f"data.val_uint8_tiles: validation tiles must be {S}x{S} (the reference centre-crops on the CPU)")
        if cfg.train.get("use_cell_metrics"):        # src/train.py:111-114
            from miphei_vit_amd.cells import CellMetrics
            if "nuclei" not in val_tiles or "slide_name" not in val_tiles:
                raise ValueError("train.use_cell_metrics needs nuclei and slide_name arrays in data.val_uint8_tiles")
            if val_tiles["nuclei"].shape[:3] != vi_.shape[:3] or len(val_tiles["slide_name"]) != vi_.shape[0]:
                raise ValueError("data.val_uint8_tiles: nuclei must be [N,H,W] and slide_name [N], matching the images")
            cell_metrics = CellMetrics(sorted(set(str(s_) for s_ in val_tiles["slide_name"])), list(cfg.data.targ_channel_names))
    module = ModelModule(generator, None, cfg.train.learning_rate_g * B ** 0.5, cfg.train.learning_rate_d, loss,
                         cell_metrics=cell_metrics, gan_train=cfg.train.gan_train).to(dev)
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
    augment = None
    if tiles is not None:
        from miphei_vit_amd.io_stage import TrainAugmenter, shuffled_indices
        # counter-based draws: sample n of the run is global (step * world * B + rank * B + b), so a rank layout change does not
        # change what a sample looks like
        augment = TrainAugmenter(dev, (S, S), seed=int(cfg.train.get("seed") or 0))
    t0 = time.perf_counter()
    for i in range(start, steps):
        if augment is not None:
            n0 = (i * world + rank) * B
            idx = shuffled_indices(n0, B, tiles[0].shape[0], int(cfg.train.get("seed") or 0), device=dev)
            batch = augment(tiles[0][idx], tiles[1][idx], n0)
        else:
            x, y = synthetic_batch(1234 + rank * 1000 + i, B, S, nc, dev)
            batch = {"image": x, "target": y}
        out = module.training_step(batch, i)
        if rank == 0 and (i % 10 == 0 or i == steps - 1):
            print(f"step {i:5d}  loss {float(out):.4f}  lr {module.current_lr(i):.3e}", flush=True)
        if rank == 0 and every and (i + 1) % every == 0 and i + 1 < steps:
            # rank 0's replica: parameters are identical on all ranks, BatchNorm running statistics are rank-local (noted in
            # the file as "bn_running_stats": "rank-local")
            save_checkpoint_atomic(module.checkpoint_state(), os.path.join(logdir, "last.ckpt"))
    module.on_train_end()           # drains the asynchronous NaN guard
    torch.cuda.synchronize()
    if rank == 0:
        # the trained weights are on disk before anything else can fail; the throughput line covers the training loop only
        dt = time.perf_counter() - t0
        print(f"{steps - start} steps, {world * B * (steps - start) / dt:.1f} tiles/s")
        save_checkpoint_atomic(module.checkpoint_state(), os.path.join(logdir, "last.ckpt"))
        save_pruned_safetensors(generator, os.path.join(logdir, "model.safetensors"))
        print("saved", os.path.join(logdir, "model.safetensors"))
    if val_tiles is not None and rank == 0:
        from miphei_vit_amd.io_stage import InputStage
        stage = InputStage(dev)
        vi, vt = torch.from_numpy(val_tiles["image"]).to(dev), torch.from_numpy(val_tiles["target"]).to(dev)
        losses_v = []
        for j0 in range(0, vi.shape[0], B):
            vb = {"image": stage.image(vi[j0:j0 + B].contiguous()), "target": stage.target(vt[j0:j0 + B].contiguous())}
            if module.use_cell_metrics:
                vb["nuclei"] = torch.from_numpy(val_tiles["nuclei"][j0:j0 + B].astype("int32")).to(dev)
                vb["slide_name"] = [str(s_) for s_ in val_tiles["slide_name"][j0:j0 + B]]
            losses_v.append(float(module.validation_step(vb, j0 // B)))
        msg = f"validation: {vi.shape[0]} tiles, val_gen_loss_sim {sum(losses_v) / len(losses_v):.4f}"
        if module.use_cell_metrics:
            n_cells = sum(int(t.numel()) for st in module.cell_metrics.state.values() for t in st["cell_id"])
            msg += f", cell_metrics: {n_cells} nuclei over {len(module.cell_metrics.state)} slides"
        print(msg, flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
