import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
import miphei_vit_amd.ops as ops
B,H,W,cin,cp,ldx,cout,rot = [int(v) for v in sys.argv[1:9]]
w = torch.randn(cout, cin, 3, 3, device="cuda") * (9*cin)**-0.5
x = torch.randn(B,H,W,cin, device="cuda").bfloat16()
xb = torch.zeros(B,H,W,ldx, device="cuda", dtype=torch.bfloat16)
perm = (torch.arange(cin, device="cuda") + rot) % cin
xb[..., :cin] = x[..., perm]
print("pack", flush=True)
wp = ops.pack_conv3x3_direct(w, cout, cp, rot=rot)
torch.cuda.synchronize(); print("packed", wp.shape, flush=True)
y = torch.zeros(B,H,W,cout, device="cuda", dtype=torch.bfloat16)
ops.conv3x3_direct(xb, wp, y, B=B,H=H,W=W,cin_pad=cp,ldx=ldx,cout=cout,ldy=cout)
torch.cuda.synchronize(); print("conv done", flush=True)
ref = F.conv2d(x.float().permute(0,3,1,2), w.bfloat16().float(), padding=1).permute(0,2,3,1)
print("rel", float((y.float()-ref).norm()/ref.norm()))
