"""Measurement only: sustained shader clock and socket power (rocm-smi) while one GEMM runs back to back for a few seconds:
the library kernel vs hipBLASLt, random vs all-zero operands.  The dense-MFMA loops on MI355X are power-limited: the same
instruction stream runs ~1.3 GHz on random bf16 data and ~1.9 GHz on zeros."""
import sys, os, subprocess, threading, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import miphei_vit_amd.ops as ops


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            p = re.search(r"Power \(W\):\s*([\d.]+)", txt)
            c = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", txt)
            out.append((float(p.group(1)) if p else None, int(c.group(1)) if c else None))
        except Exception:  # noqa: BLE001
            pass
        time.sleep(0.3)


m, n, k = 8192, 8192, 8192
for kind in ("random", "zeros"):
    a = (torch.randn(m, k, device="cuda") if kind == "random" else torch.zeros(m, k, device="cuda")).bfloat16()
    b = (torch.randn(n, k, device="cuda") if kind == "random" else torch.zeros(n, k, device="cuda")).bfloat16()
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    bt = b.t()
    for name, fn in (("ours", lambda: ops.gemm(a, b, c)), ("hipBLASLt", lambda: torch.matmul(a, bt, out=c))):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        stop, out = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, out))
        th.start()
        t0 = time.perf_counter()
        it = 0
        while time.perf_counter() - t0 < 4.0:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            it += 50
        dt = time.perf_counter() - t0
        stop.set(); th.join()
        pw = [p for p, _ in out if p]
        ck = [c for _, c in out if c]
        print(f"{kind:6s} {name:9s}: {2*m*n*k*it/dt/1e12:7.1f} TF/s sustained over {dt:.1f} s | power {sum(pw)/max(1,len(pw)):.0f} W (max {max(pw, default=0):.0f}) "
              f"| sclk samples {sorted(set(ck))[:1]}..{sorted(set(ck))[-1:]}", flush=True)
