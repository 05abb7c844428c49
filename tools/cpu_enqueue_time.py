"""Host-side cost of enqueueing one training step (no device sync inside the timed call): must stay well below the
device time of the step, otherwise the GPU idles at step boundaries."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from miphei_vit_amd.generators import get_vitmatte
from miphei_vit_amd.loss import WeightedMSELoss
from miphei_vit_amd.models import ModelModule

dev = torch.device("cuda:0")
with torch.device(dev):
    model = get_vitmatte("hoptimus0", 256, 16, use_lora=True, pretrained=False)
bench.synthetic_init_(model, seed=0)
mod = ModelModule(model, None, 8e-4, 0.0, WeightedMSELoss(50.0, torch.ones(16))).to(dev)
mod.total_iters, mod.update_pix_metrics = 100000, False
x, y = bench.synthetic_batch(1, 16, 256, 16, dev)
for i in range(4):
    mod.training_step({"image": x, "target": y}, i)
torch.cuda.synchronize()
ts = []
for i in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mod.training_step({"image": x, "target": y}, i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
print("enqueue ms / total ms per step:", [(round(a, 2), round(b, 2)) for a, b in ts])
import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
mod.training_step({"image": x, "target": y}, 9)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
