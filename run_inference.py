"""Inference entry point with the reference's shape (``/root/reference/run_inference.py`` + ``src/inference.py:87-186``):

    python run_inference.py --checkpoint_dir logs --batch_size 64 [+default_configs=miphei-vit ...]

Builds the generator from the config, loads ``model.safetensors`` / ``model.weights.ckpt`` from the checkpoint directory
(LoRA + decoder keys, ``validate_load_info`` rules), runs the hipGraph-captured forward on uint8 tiles through the
on-device input stage and writes uint8 predictions (the reference writes one TIFF per tile with pyvips; here one .npy
per batch -- image I/O is outside the path).
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint_dir", required=True)
    ap.add_argument("--batch_size", type=int, default=64)
    ap.add_argument("--num_batches", type=int, default=2)
    ap.add_argument("--output_dir", default=None)
    ap.add_argument("--allow-random-encoder", action="store_true",
                    help="run with RANDOM weights of the frozen encoder architecture when model.encoder.encoder_weights is null "
                         "(benchmarking / smoke runs only: the predictions are meaningless)")
    a, overrides = ap.parse_known_args()
    from miphei_vit_amd.checkpoint import load_generator_checkpoint
    from miphei_vit_amd.config import compose
    from miphei_vit_amd.generators import get_generator
    from miphei_vit_amd.io_stage import InputStage, export_uint8
    from miphei_vit_amd.synthetic import synthetic_init_

    cfg = compose(os.path.join(ROOT, "configs"), overrides)
    dev = torch.device("cuda", 0)
    nc, S = len(cfg.data.targ_channel_names), int(cfg.data.tile_size)
    with torch.device(dev):
        gen = get_generator(cfg.model.model_name, S, 3, nc, cfg)
    if cfg.model.encoder.encoder_weights is None:
        # The checkpoint directory holds LoRA + decoder keys only; the frozen foundation encoder comes from
        # model.encoder.encoder_weights.  The reference downloads H-Optimus-0 or fails (foundation_models.py:59-66); there is no
        # network here, so without weights this refuses unless explicitly told to produce numbers for timing.
        if not (a.allow_random_encoder or os.environ.get("MIPHEI_RANDOM_INIT") == "1"):
            raise SystemExit("model.encoder.encoder_weights is null: the frozen encoder has no weights to load.  Pass "
                             "++model.encoder.encoder_weights=/path/to/hoptimus0.safetensors, or --allow-random-encoder "
                             "(MIPHEI_RANDOM_INIT=1) for a timing run whose predictions are meaningless.")
        print("WARNING: frozen encoder initialised with RANDOM weights (--allow-random-encoder): predictions are meaningless",
              file=sys.stderr, flush=True)
        synthetic_init_(gen, seed=0)
    load_generator_checkpoint(gen, a.checkpoint_dir)
    gen.eval()
    run, x_static, out_static = gen._engine.capture_inference(a.batch_size)
    stage = InputStage(dev)
    out_dir = a.output_dir or os.path.join(a.checkpoint_dir, "predictions")
    os.makedirs(out_dir, exist_ok=True)
    g = torch.Generator(device=dev).manual_seed(0)
    for i in range(a.num_batches):
        rgb = torch.randint(0, 256, (a.batch_size, S, S, 3), generator=g, device=dev, dtype=torch.uint8)
        x_static.copy_(stage.image(rgb))
        run()
        np.save(os.path.join(out_dir, f"batch_{i:04d}.npy"), export_uint8(out_static).cpu().numpy())
    print("wrote", a.num_batches, "batches to", out_dir)


if __name__ == "__main__":
    main()
