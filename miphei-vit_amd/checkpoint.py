"""Checkpoint ingestion (load-time host logic).

Mirrors ``validate_load_info`` / ``get_generator_state_dict`` (``/root/reference/src/inference.py:28-45,79-84``), the
pruning of ``scripts/ckpt_remove_foundation_model.py:7-44`` (deployment files hold only LoRA + decoder keys) and the
loading branch of ``inference_model`` (``inference.py:135-153``).
"""
from __future__ import annotations

import os

import torch

ENCODER_PARTS = ("encoder.vit.", "encoder.model.")


def validate_load_info(load_info):
    """Raise on unexpected keys, on missing LoRA keys and on missing keys outside the frozen encoder."""
    if load_info.unexpected_keys:
        raise ValueError(f"Unexpected keys in state_dict: {load_info.unexpected_keys}")
    for key in load_info.missing_keys:
        if ".lora" in key:
            raise ValueError(f"Missing LoRA checkpoint in state_dict: {key}")
        elif not any(part in key for part in ENCODER_PARTS):
            raise ValueError(f"Missing key in state_dict: {key}")


def get_generator_state_dict(state_dict):
    return {k.replace("generator.", "", 1): v for k, v in state_dict.items() if k.startswith("generator.")}


def remove_foundation_model_ckpt(state_dict, prefix="generator."):
    """Keep LoRA adapters and everything outside the encoder (the frozen foundation weights are shipped separately)."""
    out = {}
    for k, v in state_dict.items():
        if (prefix + "encoder.vit" in k) or (prefix + "encoder.model" in k):
            if ".lora" in k:
                out[k] = v
        else:
            out[k] = v
    return out


def save_pruned_safetensors(generator, path):
    from safetensors.torch import save_file
    sd = remove_foundation_model_ckpt({k: v.detach().cpu().contiguous() for k, v in generator.state_dict().items()}, prefix="")
    save_file(sd, str(path))
    return sorted(sd)


def save_checkpoint_atomic(state, path):
    """``torch.save`` into a temporary file in the same directory, then ``os.replace``: a crash in the middle of a write leaves
    the previous resume point intact (the reference's Lightning ``ModelCheckpoint`` writes through a temporary file as well)."""
    path = str(path)
    tmp = f"{path}.tmp.{os.getpid()}"
    try:
        torch.save(state, tmp)
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return path


def load_generator_checkpoint(generator, checkpoint_dir):
    """``model.safetensors`` (LoRA + decoder, strict=False + validation) or ``model.weights.ckpt`` (Lightning)."""
    st = os.path.join(str(checkpoint_dir), "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        info = generator.load_state_dict(load_file(st), strict=False)
        validate_load_info(info)
        return info
    ck = os.path.join(str(checkpoint_dir), "model.weights.ckpt")
    sd = get_generator_state_dict(torch.load(ck, map_location="cpu", weights_only=True)["state_dict"])
    return generator.load_state_dict(sd)
