"""Is the training step dispatch-bound anywhere?  The same step captured once in a hipGraph and replayed, against eager launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from miphei_vit_amd.generators import get_vitmatte
from miphei_vit_amd.loss import WeightedMSELoss, marker_weights_from_file
from miphei_vit_amd.models import ModelModule
dev = torch.device("cuda:0")
nc, B = 16, 16
weights = marker_weights_from_file(os.path.join(bench.ROOT, "configs", "channel_stats_orion.json"), bench.ORION_MARKERS)
with torch.device(dev):
    model = get_vitmatte("hoptimus0", 256, nc, use_lora=True, pretrained=False)
bench.synthetic_init_(model, seed=0)
mod = ModelModule(model, None, 2e-4 * 4, 0., WeightedMSELoss(50.0, weights)).to(dev)
mod.total_iters = 100000
mod.update_pix_metrics = False
mod.nan_check = False
x, y = bench.synthetic_batch(1234, B, 256, nc, dev)
eng = model._engine
w = mod.loss_reconstruct.marker_weights.to(dev)
def step():
    out = eng.forward(x, train=True)
    loss, dY = eng.loss_and_grad(out, y, w, 50.0)
    getattr(eng, "backward_fused", eng.backward)(dY)
    eng.adam_step(1e-4, betas=(0.5, 0.999), eps=1e-7, max_norm=1.0)
for _ in range(5): step()
torch.cuda.synchronize()
def timeit(fn, n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"eager  : {timeit(step):.3f} ms/step")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    step()
print(f"graph  : {timeit(g.replay):.3f} ms/step")
print(f"eager  : {timeit(step):.3f} ms/step")
