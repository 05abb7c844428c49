"""Which gradients / buffers of the full-size training step differ between two runs in MIPHEI_DETERMINISTIC=1?
Runs forward + loss + backward twice in one process from the same weights and inputs and compares, bit for bit, the output,
every named parameter's slice of the flat gradient buffer, the BatchNorm batch statistics and the saved activations.
    MIPHEI_DETERMINISTIC=1 python tools/debug/det_bisect.py [batch] [encoder]"""
import os, sys
os.environ.setdefault("MIPHEI_DETERMINISTIC", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from miphei_vit_amd import ops
from miphei_vit_amd.generators import get_vitmatte

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
enc = sys.argv[2] if len(sys.argv) > 2 else "hoptimus0"
nc, img = 16, 256
dev = torch.device("cuda:0")
with torch.device(dev):
    model = get_vitmatte(enc, img, nc, use_lora=True, pretrained=False)
bench.synthetic_init_(model, seed=13)
model.train()
eng = model._engine
x, y = bench.synthetic_batch(300, B, img, nc, dev)
w = torch.ones(nc, device=dev)
print("deterministic mode:", ops.DETERMINISTIC)


def snap(t):
    return None if t is None else t.detach().clone()


def run():
    out = eng.forward(x, train=True, bn_train=True)
    loss, dY = eng.loss_and_grad(out, y, w, 50.0)
    ws = eng._saved.w
    fwd = {"out": snap(out), "feat": snap(ws.feat)}
    for j, t in enumerate(ws.pre_f):
        fwd[f"pre_f{j}"] = snap(t)
    for j, t in enumerate(ws.pre_c):
        fwd[f"pre_c{j}"] = snap(t)
    for j in (0, 20, 39):
        if j < len(ws.qkv):
            fwd[f"qkv{j}"], fwd[f"o{j}"] = snap(ws.qkv[j]), snap(ws.o[j])
    g = eng.backward(dY)
    torch.cuda.synchronize()
    bwd = {}
    fl = eng._ensure_flat()
    for name, p in model.named_parameters():
        if p.requires_grad and id(p) in fl.gview:
            bwd[name] = snap(fl.gview[id(p)])
    for k in ("dcat", "dpre_f", "dFpost"):
        v = getattr(ws, k, None)
        if isinstance(v, (list, tuple)):
            for j, t in enumerate(v):
                if torch.is_tensor(t):
                    bwd[f"ws.{k}{j}"] = snap(t)
    for k in ("dx", "dy", "dh", "do", "du"):
        v = getattr(ws, k, None)
        if torch.is_tensor(v):
            bwd[f"ws.{k}(block 0)"] = snap(v)
    return fwd, bwd, float(loss)


a_f, a_b, la = run()
b_f, b_b, lb = run()
print("loss", la, lb)
nd = 0
for tag, A, Bd in (("fwd", a_f, b_f), ("bwd", a_b, b_b)):
    for k in A:
        if A[k] is None:
            continue
        same = torch.equal(A[k], Bd[k])
        if not same:
            nd += 1
            d = (A[k].double() - Bd[k].double()).abs().max().item()
            nz = int((A[k] != Bd[k]).sum())
            if nd <= 60:
                print(f"DIFF {tag} {k:60s} max|d| {d:.3e}  differing elements {nz} / {A[k].numel()}")
print("differing tensors:", nd)
