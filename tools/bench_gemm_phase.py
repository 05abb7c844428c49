import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import miphei_vit_amd.ops as ops
M = 16 * 329
for (m, n, k) in [(M, 4608, 1536), (M, 1536, 8192), (M, 1536, 1536)]:
    a = torch.randn(m, k, device="cuda").bfloat16(); b = torch.randn(n, k, device="cuda").bfloat16()
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    st = torch.zeros(256 * 4, device="cuda", dtype=torch.float64)
    for _ in range(3): ops.gemm(a, b, c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm(a, b, c, stats=st, nslots=1, flags=0x4000); e1.record(); torch.cuda.synchronize()
    s = st.view(256, 4).cpu()
    tiles = s[:, 2].sum()
    print(f"M={m} N={n} K={k}: kernel {e0.elapsed_time(e1)*1e3:.1f} us; per tile: mainloop {float(s[:,0].sum()/tiles):.0f} clk, "
          f"epilogue(+next prologue issue) {float(s[:,1].sum()/tiles):.0f} clk; tiles/block {float(tiles/256):.2f}; "
          f"per-block total {float((s[:,0]+s[:,1]).max()):.0f} clk (max) {float((s[:,0]+s[:,1]).mean()):.0f} (mean)")
