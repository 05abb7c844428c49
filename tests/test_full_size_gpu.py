"""Parity at BASELINE.json's full model size (configs[1]: H-Optimus-0 ViT-g/14 + LoRA + ViTMatte decoder, 256x256, 16 markers):
the 1.14 B-parameter HIP path against the fp32 CPU oracle on one synthetic tile pair - outputs within the north-star tolerance
(1e-3 relative MSE per channel) and the training loss / gradient norm of the 6.7 M trainable parameters."""
import pytest
import torch

pytestmark = pytest.mark.gpu


# (16, 256) = BASELINE.json configs[1] itself: 256-row GEMM tiles, M = 5264 (about 40 s of CPU oracle);
# (2, 512) = the shape of configs[3]: 1301 tokens, 36 -> 32 regrid, decoder at 512 x 512
@pytest.mark.parametrize("B,img", [(2, 256), (16, 256), (2, 512)])
def test_hoptimus0_forward_loss_gradnorm_vs_oracle(B, img):
    import bench
    from oracle import VIT_CONFIGS
    from oracle.model import OracleTrainer, orion_marker_weights
    from miphei_vit_amd.generators import get_vitmatte
    from miphei_vit_amd.loss import WeightedMSELoss
    nc = 16
    dev = torch.device("cuda:0")
    with torch.device(dev):
        model = get_vitmatte("hoptimus0", img, nc, use_lora=True, pretrained=False)
    bench.synthetic_init_(model, seed=3)
    if img == 256:
        assert sum(p.numel() for p in model.parameters()) == 1_141_576_432       # SURVEY.md section 8 a1 [probe]
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 6_697_712
    x, y = bench.synthetic_batch(77, B, img, nc, dev)
    w = orion_marker_weights(nc)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    p = {k: v.detach().to("cpu", torch.float32) for k, v in model.state_dict().items()}
    tr = OracleTrainer(p, VIT_CONFIGS["hoptimus0"], nc, batch_size=B, total_iters=1000, weights=w)
    tr.p = p
    out_ref, loss_ref, gref = tr.loss_and_grads(x.cpu(), y.cpu())
    model.train()
    out = model(x)
    loss = WeightedMSELoss(50.0, w).to(dev)(y_true=y, y_pred=out)
    loss.backward()
    o = out.detach().float().cpu()
    rel = ((o - out_ref) ** 2).sum(dim=(0, 2, 3)) / (out_ref ** 2).sum(dim=(0, 2, 3))
    assert float(rel.max()) < 1e-3, rel
    assert abs(float(loss.detach()) - float(loss_ref)) < 2e-3 * abs(float(loss_ref))
    named = dict(model.named_parameters())
    gn_hip = torch.cat([named[k].grad.flatten().double().cpu() for k in gref]).norm()
    gn_ref = torch.cat([g.flatten().double() for g in gref.values()]).norm()
    assert abs(float(gn_hip) - float(gn_ref)) < 0.05 * float(gn_ref), (float(gn_hip), float(gn_ref))
    # the shallow end of the backward pass (heads, last fusion block) is only a few bf16 roundings away from fp32
    for k in ("decoder.segmentation_head_3.1.weight", "decoder.segmentation_head_0.0.psi.3.weight", "decoder.fusion_blks.3.conv.bn.weight"):
        g1, g0 = named[k].grad.double().cpu(), gref[k].double()
        assert float((g1 - g0).norm() / g0.norm()) < 0.05, k


def test_hoptimus0_batch16_tiles_agree_with_batch2_chunks():
    """The batch-16 step runs the 256-row GEMM tiles (M = 5264) and three attention row blocks per pair; the oracle-checked
    batch-2 run above runs the 128-row tiles (M = 658).  In eval mode (running BatchNorm statistics) a tile's prediction does
    not depend on its batch, so the two paths must agree tile by tile, to bf16 rounding."""
    import bench
    from miphei_vit_amd.generators import get_vitmatte
    nc, img = 16, 256
    dev = torch.device("cuda:0")
    with torch.device(dev):
        model = get_vitmatte("hoptimus0", img, nc, use_lora=True, pretrained=False)
    bench.synthetic_init_(model, seed=5)
    model.eval()
    x, _ = bench.synthetic_batch(11, 16, img, nc, dev)
    with torch.no_grad():
        full = model(x).float()
        parts = torch.cat([model(x[i:i + 2]).float() for i in range(0, 16, 2)])
    rel = ((full - parts) ** 2).sum(dim=(0, 2, 3)) / (parts ** 2).sum(dim=(0, 2, 3))
    assert float(rel.max()) < 2e-4, rel     # both are bf16 paths: different tile shapes = different summation order only
