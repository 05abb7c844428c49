"""Fixture for the on-device input stage from the REFERENCE's own ``NormalizationLayer`` / ``get_input_mean_std``
(/root/reference/src/dataset.py:545-575, 593-606), build container only.  dataset.py is imported by file path with stub modules
for what the container lacks (albumentations, pyvips, slidevips, skimage: none of them is touched by the two functions used).

  comp_io.npz : uint8 H&E tile [40, 52, 3] and mIF tile [40, 52, 16] (numpy RNG, stored), NormalizationLayer(mode="he") with the
                mean / std get_input_mean_std returns for model_name "myvitmatte" + encoder "hoptimus0", NormalizationLayer(mode=
                "if"), and unormalize of both.
Usage:  python oracle/make_golden_io.py
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.make_golden import REF, _exec  # noqa: E402


def load_dataset_module():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class ImageOnlyTransform:                       # base class of the reference's colour augmentors (never instantiated here)
        def __init__(self, *a, **k):
            pass

    alb = stub("albumentations")
    stub("albumentations.core")
    stub("albumentations.core.transforms_interface", ImageOnlyTransform=ImageOnlyTransform)
    alb.core = sys.modules["albumentations.core"]
    stub("pyvips")
    stub("slidevips", SlideVips=object)
    if "skimage" not in sys.modules:
        try:
            import skimage  # noqa: F401
        except Exception:  # noqa: BLE001
            stub("skimage")
    if "refsrc" not in sys.modules:
        pkg = types.ModuleType("refsrc")
        pkg.__path__ = [f"{REF}/src"]
        sys.modules["refsrc"] = pkg
    _exec("refsrc", "augmentations", f"{REF}/src/augmentations.py")
    return _exec("refsrc", "dataset", f"{REF}/src/dataset.py")


def main():
    ds = load_dataset_module()
    rng = np.random.default_rng(2024)
    rgb = rng.integers(0, 256, size=(40, 52, 3), dtype=np.uint8)
    rgb[0, :4] = [[0, 0, 0], [255, 255, 255], [1, 254, 127], [128, 3, 77]]
    mif = rng.integers(0, 256, size=(40, 52, 16), dtype=np.uint8)
    mif[0, 0, :4] = [0, 255, 1, 254]
    cfg = types.SimpleNamespace(model=types.SimpleNamespace(model_name="myvitmatte_hoptimus0_lora",
                                                            encoder=types.SimpleNamespace(encoder_name="hoptimus0")))
    stats = ds.get_input_mean_std(cfg, None)
    he = ds.NormalizationLayer(stats, mode="he")
    mf = ds.NormalizationLayer(stats, mode="if")
    x, y = he(rgb), mf(mif)
    assert x.dtype == np.float32 and y.dtype == np.float32
    out = dict(rgb=rgb, mif=mif, mean=np.asarray(stats["mean"], dtype=np.float64), std=np.asarray(stats["std"], dtype=np.float64),
               he=x, mif_norm=y, he_unorm=np.asarray(he.unormalize(x), dtype=np.float32),
               if_unorm=np.asarray(mf.unormalize(y), dtype=np.float32))
    path = os.path.join(ROOT, "tests", "golden", "comp_io.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == "__main__":
    main()
