"""Soak test of the training step's kernels for run-to-run glitches (the class of the round-4 attention bug: a rare wrong result that no
tolerance test sees).  MIPHEI_DETERMINISTIC=1: forward + loss + backward of the benchmark configuration (H-Optimus-0, B = 16) repeated N
times on ONE input with unchanged weights; the output tensor and the flat gradient buffer of every repeat are compared bit for bit with
the first.  Prints the number of differing repeats (and, for the first one, which slices of the gradient buffer differ).
  MIPHEI_DETERMINISTIC=1 python tools/debug/step_soak.py [N=30] [B=16]"""
import hashlib, os, sys
os.environ.setdefault("MIPHEI_DETERMINISTIC", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from oracle.model import orion_marker_weights
from miphei_vit_amd import _lib, ops
if os.environ.get("MIPHEI_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIPHEI_LIB"])
from miphei_vit_amd.generators import get_vitmatte
from miphei_vit_amd.loss import WeightedMSELoss

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
nc, img = 16, 256
dev = torch.device("cuda:0")
assert ops.DETERMINISTIC, "run with MIPHEI_DETERMINISTIC=1"
with torch.device(dev):
    model = get_vitmatte("hoptimus0", img, nc, use_lora=True, pretrained=False)
bench.synthetic_init_(model, seed=13)
model.to(dev).train()
eng = model._engine
loss_fn = WeightedMSELoss(50.0, orion_marker_weights(nc)).to(dev)
x, y = bench.synthetic_batch(300, B, img, nc, dev)


def once():
    out = eng.forward(x, train=True)
    loss, dY = eng.loss_and_grad(out, y, loss_fn.marker_weights, loss_fn.lambda_factor)
    getattr(eng, "backward_fused", eng.backward)(dY)
    torch.cuda.synchronize()
    return out.clone(), eng._flat.gflat.clone(), float(loss)


out0, g0, l0 = once()
bad = 0
for i in range(N):
    out, g, l = once()
    same_o, same_g = torch.equal(out, out0), torch.equal(g, g0)
    if not (same_o and same_g):
        bad += 1
        if bad == 1:
            d = (g != g0).nonzero().flatten()
            print(f"repeat {i}: output {'same' if same_o else 'DIFFERS'}, gradient elements differing {d.numel()}"
                  f" (first {d[:4].tolist()}, last {d[-4:].tolist()} of {g.numel()}; LoRA slice ends at {eng._flat.n_lora})", flush=True)
print(f"{bad} of {N} repeats differ from the first (loss {l0:.6f}, {hashlib.sha256(g0.cpu().numpy().tobytes()).hexdigest()[:16]})")
