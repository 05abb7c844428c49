"""How far ahead of the GPU the host runs in the training step: wall time of each training_step CALL (enqueue only, no
synchronisation) next to the GPU time of the step.   python tools/debug/host_ahead.py"""
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
batches = [bench.synthetic_batch(1234 + i, B, 256, nc, dev) for i in range(4)]
for i in range(6):
    mod.training_step({"image": batches[i % 4][0], "target": batches[i % 4][1]}, i)
torch.cuda.synchronize()
N = 12
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
host = []
ev[0].record()
t_start = time.perf_counter()
for i in range(N):
    t0 = time.perf_counter()
    mod.training_step({"image": batches[i % 4][0], "target": batches[i % 4][1]}, i)
    host.append((time.perf_counter() - t0) * 1e3)
    ev[i + 1].record()
t_enq = (time.perf_counter() - t_start) * 1e3
torch.cuda.synchronize()
t_all = (time.perf_counter() - t_start) * 1e3
gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
print("host ms per call:", " ".join(f"{h:.1f}" for h in host))
print("gpu  ms per step:", " ".join(f"{g:.1f}" for g in gpu))
print(f"enqueue of {N} steps {t_enq:.1f} ms, all done {t_all:.1f} ms")
