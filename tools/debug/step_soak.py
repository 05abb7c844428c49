"""Soak test of the training step's kernels for run-to-run glitches (the class of the round-4 attention bug: a rare wrong result that no
tolerance test sees).  MIPHEI_DETERMINISTIC=1: forward + loss + backward of the benchmark configuration (H-Optimus-0, B = 16) repeated N
times on ONE input with unchanged weights; the output tensor and the flat gradient buffer of every repeat are compared bit for bit with
the first.  Prints the number of differing repeats (and, for the first one, which slices of the gradient buffer differ).
  MIPHEI_DETERMINISTIC=1 python tools/debug/step_soak.py [N=30] [B=16] [img=256] [generator=myvitmatte|unet_lora] [mode=train|infer]
(infer: the eval-mode forward alone -- the kernels' inference variants: no saved pre-activations, no attention residual)
Environment: SOAK_ENCODER=tiny_swiglu|tiny4_swiglu|... (oracle.VIT_CONFIGS name, default hoptimus0), SOAK_CHUNKED=0 (fusion convolutions on the
implicit GEMM), SOAK_ATTN_RES=0 (no rounding residual of O)."""
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
img = int(sys.argv[3]) if len(sys.argv) > 3 else 256
gen = sys.argv[4] if len(sys.argv) > 4 else "myvitmatte"
infer = len(sys.argv) > 5 and sys.argv[5] == "infer"
nc = 16
dev = torch.device("cuda:0")
assert ops.DETERMINISTIC, "run with MIPHEI_DETERMINISTIC=1"
with torch.device(dev):
    if gen == "unet_lora":
        from miphei_vit_amd.generators.unet import Unet
        model = Unet(img, os.environ.get("SOAK_ENCODER", "hoptimus0"), use_lora=True, classes=nc, pretrained=False)
    else:
        model = get_vitmatte(os.environ.get("SOAK_ENCODER", "hoptimus0"), img, nc, use_lora=True, pretrained=False)
bench.synthetic_init_(model, seed=13)
model.to(dev).train()
if len(sys.argv) > 5 and sys.argv[5] == "infer":
    model.eval()
eng = model._engine
if hasattr(eng, "use_chunked_conv"):
    eng.use_chunked_conv = os.environ.get("SOAK_CHUNKED", "1") == "1"
    eng.attn_residual = os.environ.get("SOAK_ATTN_RES", "1") == "1"
loss_fn = WeightedMSELoss(50.0, orion_marker_weights(nc)).to(dev)
x, y = bench.synthetic_batch(300, B, img, nc, dev)


params = [p for p in model.parameters() if p.requires_grad]


def once():
    if infer:
        with torch.no_grad():
            out = model(x) if gen == "unet_lora" else eng.forward(x, train=False, bn_train=False)
        torch.cuda.synchronize()
        return out.clone(), torch.zeros(1, device=dev), 0.0
    if hasattr(eng, "_drop_step"):
        eng._drop_step = 0                      # UNETR: the dropout masks are functions of (seed, step): the same masks every repeat
    out = eng.forward(x, train=True)
    loss, dY = eng.loss_and_grad(out, y, loss_fn.marker_weights, loss_fn.lambda_factor)
    getattr(eng, "backward_fused", eng.backward)(dY)
    torch.cuda.synchronize()
    return out.clone(), torch.cat([p.grad.reshape(-1) for p in params]), float(loss)


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
                  f" (first {d[:4].tolist()}, last {d[-4:].tolist()} of {g.numel()}, parameters in model.parameters() order)", flush=True)
print(f"{bad} of {N} repeats differ from the first (loss {l0:.6f}, {hashlib.sha256(g0.cpu().numpy().tobytes()).hexdigest()[:16]})")
