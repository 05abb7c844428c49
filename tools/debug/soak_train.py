"""Soak run: N fused training steps of the full configuration on a fixed set of synthetic batches with the asynchronous NaN guard
active; prints the loss every 50 steps and the sustained rate."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from miphei_vit_amd.generators import get_vitmatte
from miphei_vit_amd.loss import WeightedMSELoss
from miphei_vit_amd.models import ModelModule
from oracle.model import orion_marker_weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
with torch.device(dev):
    model = get_vitmatte("hoptimus0", 256, 16, use_lora=True, pretrained=False)
bench.synthetic_init_(model, seed=0)
mod = ModelModule(model, None, 2e-4 * 4, 0., WeightedMSELoss(50.0, orion_marker_weights(16))).to(dev)
mod.total_iters = 2000
batches = [bench.synthetic_batch(100 + i, 16, 256, 16, dev) for i in range(8)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    x, y = batches[i % 8]
    loss = mod.training_step({"image": x, "target": y}, i)
    if i % 50 == 0 or i == n - 1:
        print(f"step {i:4d} lr {mod.current_lr():.2e} loss {float(loss):.4f}", flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{n} steps, {16 * n / dt:.1f} tiles/s sustained (includes the loss read-backs above)")
