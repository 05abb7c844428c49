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
    """The MIPHEI-ViT branch (``myvitmatte*``) and the UNETR baseline on the registry ViTs (``unet*``); smp_unet / hemit are not."""
    if model_name.startswith("myvitmatte"):
        if nc_in != 3:
            raise ValueError("MIPHEI-ViT takes 3-channel H&E tiles")
        ckpt_path = _cfg_get(cfg, "model.encoder.encoder_weights")
        pretrained = _cfg_get(cfg, "model.encoder.pretrained", True)
        return get_vitmatte(_cfg_get(cfg, "model.encoder.encoder_name"), img_size, nc_out, use_lora=True,
                            ckpt_path=ckpt_path, pretrained=pretrained)
    if model_name.startswith("unet"):
        # UNETR baseline (reference src/generators/__init__.py:25-41): forward, backward, Dropout / DropPath
        if _cfg_get(cfg, "train.foreground_head", False):
            raise NotImplementedError
        from .unet import Unet
        gen = Unet(img_size=img_size, encoder_name=_cfg_get(cfg, "model.encoder.encoder_name"),
                   encoder_weights=_cfg_get(cfg, "model.encoder.encoder_weights"), decoder_out_channels=32,
                   head_use_attention=True, use_lora="lora" in model_name, classes=nc_out,
                   drop_rate=_cfg_get(cfg, "model.dropout", 0.0) or 0.0, pretrained=_cfg_get(cfg, "model.encoder.pretrained", True))
        if "frozen" in model_name:
            gen.freeze_encoder()
        return gen
    raise NotImplementedError(f"generator '{model_name}' is outside the MI355X hot path ('myvitmatte*', 'unet*')")
