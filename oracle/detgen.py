"""Deterministic, torch-RNG-free tensor generator (test infrastructure).

A counter-based generator: value i of stream (seed, name) is
splitmix64(key(seed, name) + i) mapped to (0,1) and, for normals, through
Box-Muller.  Pure numpy integer/float arithmetic, so the build container and
the GPU box regenerate bit-identical weights and inputs without shipping
tensors and without depending on torch's RNG streams.
"""

from __future__ import annotations

import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def _key(seed: int, name: str) -> np.uint64:
    h = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    k = (int(seed) & 0xFFFFFFFF) << 32 | h
    return _splitmix64(np.array([k], dtype=np.uint64))[0]


def det_uniform(seed: int, name: str, shape) -> np.ndarray:
    """float64 uniforms in (0, 1), shape `shape`."""
    n = int(np.prod(shape)) if len(tuple(shape)) else 1
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(2) + _key(seed, name)
    bits = _splitmix64(ctr) >> np.uint64(11)  # 53 bits
    u = (bits.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    return u.reshape(shape)


def det_normal(seed: int, name: str, shape, mean: float = 0.0, std: float = 1.0) -> np.ndarray:
    """float32 normals N(mean, std) via Box-Muller on two hashed uniforms."""
    n = int(np.prod(shape)) if len(tuple(shape)) else 1
    with np.errstate(over="ignore"):
        base = np.arange(n, dtype=np.uint64) * np.uint64(2) + _key(seed, name)
        b1 = _splitmix64(base) >> np.uint64(11)
        b2 = _splitmix64(base + np.uint64(1)) >> np.uint64(11)
    u1 = (b1.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    u2 = (b2.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (mean + std * z).astype(np.float32).reshape(shape)


def det_state_dict(shapes: dict, seed: int = 0, layerscale: float = 0.5) -> dict:
    """Deterministic weights for a generator state dict.

    `shapes` maps reference state-dict keys to shapes.
    Rules: conv weights N(0, 0.02); linear weights N(0, 1/sqrt(fan_in));
    biases N(0, 0.02); LayerNorm/BatchNorm weight N(1, 0.02), bias N(0, 0.02);
    LayerScale gamma = `layerscale` (NOT timm's 1e-5, which would make the
    encoder numerically invisible); running_mean N(0, 0.1), running_var
    U(0.5, 1.5); LoRA A ~ N(0, 1/8), B ~ N(0, 0.02) (non-zero so the LoRA path
    is live); tokens / pos_embed N(0, 0.02).
    """
    out = {}
    for k, shp in shapes.items():
        shp = tuple(shp)
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[k] = np.zeros((), dtype=np.int64)
        elif leaf == "running_mean":
            out[k] = det_normal(seed, k, shp, 0.0, 0.1)
        elif leaf == "running_var":
            out[k] = (0.5 + det_uniform(seed, k, shp)).astype(np.float32)
        elif leaf == "gamma":
            out[k] = np.full(shp, layerscale, dtype=np.float32)
        elif leaf == "A":
            out[k] = det_normal(seed, k, shp, 0.0, 1.0 / shp[1] ** 0.5)  # randn/sqrt(rank), lora.py:11-12
        elif leaf == "B":
            out[k] = det_normal(seed, k, shp, 0.0, 0.02)
        elif leaf in ("cls_token", "reg_token", "pos_embed"):
            out[k] = det_normal(seed, k, shp, 0.0, 0.02)
        elif leaf == "weight":
            if len(shp) == 4:  # conv (incl. patch_embed.proj)
                if "patch_embed" in k:
                    fan_in = shp[1] * shp[2] * shp[3]
                    out[k] = det_normal(seed, k, shp, 0.0, 1.0 / fan_in ** 0.5)
                else:
                    out[k] = det_normal(seed, k, shp, 0.0, 0.02)
            elif len(shp) == 2:  # linear
                out[k] = det_normal(seed, k, shp, 0.0, 1.0 / shp[1] ** 0.5)
            else:  # norm weight
                out[k] = det_normal(seed, k, shp, 1.0, 0.02)
        elif leaf == "bias":
            out[k] = det_normal(seed, k, shp, 0.0, 0.02)
        else:
            raise KeyError(f"det_state_dict: no rule for {k}")
    return out
