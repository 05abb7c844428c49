"""Where does the encoder FORWARD of the full-size model stop being run-to-run identical?  Runs the training forward N times in one
process (same weights, same input) and compares bitwise checksums of every block's saved tensors against run 0.
    python tools/debug/det_fwd.py [batch] [runs]         (MVIT_GEMM_WS=0 with MIPHEI_DBG_LIB=1 switches the GEMM kernel)"""
import os, sys
os.environ.setdefault("MIPHEI_DETERMINISTIC", "1")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from miphei_vit_amd import _lib
if os.environ.get("MIPHEI_DBG_LIB") == "1":
    _lib.LIB_PATH = _lib.DBG_LIB_PATH
import torch
import bench
from miphei_vit_amd.generators import get_vitmatte

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nc, img = 16, 256
dev = torch.device("cuda:0")
with torch.device(dev):
    model = get_vitmatte("hoptimus0", img, nc, use_lora=True, pretrained=False)
bench.synthetic_init_(model, seed=13)
model.train()
eng = model._engine
x, y = bench.synthetic_batch(300, B, img, nc, dev)


def csum(t):
    v = t.detach().contiguous().view(torch.int32 if t.element_size() == 4 else torch.int16).to(torch.int64)
    return int(v.sum()), int((v * (torch.arange(v.numel(), device=v.device) % 8191).view(v.shape)).sum())


def snap():
    ws = eng._saved.w
    out = {}
    L = len(ws.qkv)
    for l in range(L):
        out[(l, "0 x_in")] = csum(ws.x_in[l])
        out[(l, "1 h1")] = csum(ws.h1[l])
        out[(l, "2 t")] = csum(ws.t[l])
        out[(l, "3 qkv")] = csum(ws.qkv[l])
        out[(l, "4 o")] = csum(ws.o[l])
        out[(l, "5 x_mid")] = csum(ws.x_mid[l])
        out[(l, "6 u")] = csum(ws.u[l])
    out[(L, "0 x_in")] = csum(ws.x_in[L])
    return out


ref = None
for r in range(runs):
    eng.forward(x, train=True, bn_train=True)
    torch.cuda.synchronize()
    s = snap()
    if ref is None:
        ref = s
        continue
    bad = sorted(k for k in s if s[k] != ref[k])
    print(f"run {r}: {len(bad)} of {len(s)} tensors differ from run 0; first: {bad[:6]}")
