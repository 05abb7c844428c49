"""Minimal composer for the reference's Hydra-style command line (Hydra / OmegaConf are not dependencies here).

Supports what ``run.py +default_configs=miphei-vit ++train.epochs=100`` needs: base ``config.yaml``, ``+group=name``
overlays (``configs/<group>/<name>.yaml`` merged over the base) and ``++a.b.c=value`` / ``a.b.c=value`` overrides.
The result is a nested mapping with attribute access (``cfg.model.encoder.encoder_name``).
"""
from __future__ import annotations

import os

import yaml


class Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def get_path(self, path, default=None):
        cur = self
        for key in path.split("."):
            if not isinstance(cur, dict) or key not in cur:
                return default
            cur = cur[key]
        return cur


def _wrap(o):
    if isinstance(o, dict):
        return Cfg({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_wrap(v) for v in o]
    return o


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def compose(config_dir, overrides=()):
    with open(os.path.join(config_dir, "config.yaml")) as f:
        cfg = yaml.safe_load(f) or {}
    sets = []
    for ov in overrides:
        if ov.startswith("+") and not ov.startswith("++"):
            group, name = ov[1:].split("=", 1)
            with open(os.path.join(config_dir, group, name + ".yaml")) as f:
                _merge(cfg, yaml.safe_load(f) or {})
        else:
            sets.append(ov.lstrip("+"))
    for ov in sets:
        key, val = ov.split("=", 1)
        cur = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            cur = cur.setdefault(p, {})
        cur[parts[-1]] = yaml.safe_load(val)
    return _wrap(cfg)
