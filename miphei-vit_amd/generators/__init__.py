"""``get_generator`` -- the drop-in boundary (reference ``src/generators/__init__.py:9-56``)."""
from .foundation_models import FOUNDATION_MODEL_REGISTRY  # noqa: F401
from .mipheivit import get_vitmatte  # noqa: F401


def _cfg_get(cfg, path, default=None):
    cur = cfg
    for key in path.split("."):
        if cur is None:
            return default
        if isinstance(cur, dict):
            cur = cur.get(key, default if key == path.split(".")[-1] else None)
        else:
            cur = getattr(cur, key, default if key == path.split(".")[-1] else None)
    return cur


def get_generator(model_name, img_size, nc_in, nc_out, cfg):
    """Only the MIPHEI-ViT branch (``myvitmatte*``) is in scope; the baselines (smp_unet, unet, hemit) are not."""
    if model_name.startswith("myvitmatte"):
        if nc_in != 3:
            raise ValueError("MIPHEI-ViT takes 3-channel H&E tiles")
        ckpt_path = _cfg_get(cfg, "model.encoder.encoder_weights")
        pretrained = _cfg_get(cfg, "model.encoder.pretrained", True)
        return get_vitmatte(_cfg_get(cfg, "model.encoder.encoder_name"), img_size, nc_out, use_lora=True,
                            ckpt_path=ckpt_path, pretrained=pretrained)
    raise NotImplementedError(f"generator '{model_name}' is outside the MI355X hot path (only 'myvitmatte*')")
