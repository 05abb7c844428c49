"""Synthetic weights and tiles of the benchmark / smoke configuration: random parameters of the
H-Optimus-0 + MIPHEI decoder architecture and H&E-like / mIF-like tile batches generated on the device.  Used by ``run.py`` /
``run_inference.py`` when no encoder weights are configured and by ``bench.py`` (there is no network for checkpoints or datasets)."""
import torch


def synthetic_init_(model, seed):
    """Random weights of the H-Optimus-0 + MIPHEI decoder architecture: linears N(0,1/sqrt(fan_in)),
    convs N(0,0.02), LayerScale 0.5, LoRA A~N(0,1/8) B~N(0,0.02) (live adapters), norms ~ N(1,0.02)/N(0,0.02)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            leaf = name.rsplit(".", 1)[-1]
            if leaf == "gamma":
                p.fill_(0.5)
            elif leaf == "A":
                p.copy_(torch.randn(p.shape, generator=g, device="cuda") / p.shape[1] ** 0.5)
            elif leaf == "B":
                p.copy_(torch.randn(p.shape, generator=g, device="cuda") * 0.02)
            elif leaf in ("cls_token", "reg_token", "pos_embed"):
                p.copy_(torch.randn(p.shape, generator=g, device="cuda") * 0.02)
            elif leaf == "weight" and p.dim() == 2:
                p.copy_(torch.randn(p.shape, generator=g, device="cuda") / p.shape[1] ** 0.5)
            elif leaf == "weight" and p.dim() == 4:
                if "patch_embed" in name:
                    p.copy_(torch.randn(p.shape, generator=g, device="cuda") / (p.shape[1] * p.shape[2] * p.shape[3]) ** 0.5)
                else:
                    p.copy_(torch.randn(p.shape, generator=g, device="cuda") * 0.02)
            elif leaf == "weight":
                p.copy_(1.0 + torch.randn(p.shape, generator=g, device="cuda") * 0.02)
            elif leaf == "bias":
                p.copy_(torch.randn(p.shape, generator=g, device="cuda") * 0.02)


def synthetic_batch(seed, B, S, nc, device):
    """H&E-like normalised image and mIF-like target, generated on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    mu = torch.tensor([211.1, 194.7, 213.8], device=device).view(1, 3, 1, 1)
    sd = torch.tensor([30.1, 36.4, 26.4], device=device).view(1, 3, 1, 1)
    rgb = (mu + sd * torch.randn(B, 3, S, S, generator=g, device=device)).round().clamp(0, 255)
    mean = torch.tensor([0.707223, 0.578729, 0.703617], device=device).view(1, 3, 1, 1) * 255
    std = torch.tensor([0.211883, 0.230117, 0.177517], device=device).view(1, 3, 1, 1) * 255
    image = (rgb - mean) / std
    u = torch.rand(B, nc, S, S, generator=g, device=device).clamp_min(1e-12)
    t8 = (-20.0 * u.log()).floor().clamp(max=255)
    target = t8 / 255.0 * 1.8 - 0.9
    return image.contiguous(), target.contiguous()
