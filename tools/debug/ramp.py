"""What the FIRST timed step of bench.py pays (2-4 ms, `unprobed.step_ms[0]`): 400 launches of the qkv GEMM timed one by one right after a
synchronisation followed by 0 / 1 / 20 ms of idle.  After any idle the first launch takes 110-230 us, launches 5-20 run at boost (67 us), 20-50
at 77-80 us and 100-200 at 72-73 us before the steady 68.5 us returns: a ~14 ms transient of the power controller, ~1 ms of lost time for this
kernel alone -- not host launch overhead (HIP_FORCE_DEV_KERNARG=1 changes nothing) and not kernel work (tools/debug/first_step.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import miphei_vit_amd.ops as ops
M, D = 5264, 1536
a = torch.randn(M, D, device="cuda").bfloat16(); w = torch.randn(3 * D, D, device="cuda").bfloat16(); c = torch.empty(M, 3 * D, device="cuda", dtype=torch.bfloat16)
for _ in range(200): ops.gemm(a, w, c)
torch.cuda.synchronize()
for idle_ms in (0, 1, 20):
    time.sleep(idle_ms / 1e3)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(401)]
    ev[0].record()
    for i in range(400):
        ops.gemm(a, w, c)
        ev[i + 1].record()
    torch.cuda.synchronize()
    t = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(400)]
    print(f"idle {idle_ms} ms: first 5 {[round(x) for x in t[:5]]} us; launches 5-20 avg {sum(t[5:20])/15:.1f}; 20-50 avg {sum(t[20:50])/30:.1f}; 100-200 avg {sum(t[100:200])/100:.1f}; 300-400 avg {sum(t[300:400])/100:.1f}")
