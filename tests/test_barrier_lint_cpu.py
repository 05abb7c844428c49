"""Source lint (CPU): every raw `s_barrier` in the HIP sources either has an `s_waitcnt ... lgkmcnt(0)` within the ten lines in front of it or is
marked `[no LDS reads pending]` with the reason.

`s_barrier` does not wait for LDS reads in flight, and hipcc may sink the MFMAs that consume a wave's last fragment reads below the barrier (they
are not memory operations): the reads then cross it unfinished while a neighbour wave refills the buffer by DMA.  Round 4 found the attention
kernels returning a slightly wrong 32-query slab once in 1e2..1e4 launches that way.  `__syncthreads()` carries the waits itself."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "miphei-vit_amd", "csrc")


def test_every_raw_barrier_waits_for_lds_reads_or_says_why_not():
    bad, seen = [], 0
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp"))):
        lines = open(path).read().split("\n")
        for i, line in enumerate(lines):
            code = line.split("//")[0]
            if "__builtin_amdgcn_s_barrier()" not in code and not re.search(r'asm[^;]*"s_barrier', code):
                continue
            seen += 1
            window = "\n".join(lines[max(0, i - 10):i + 1])
            if "lgkmcnt(0)" not in window and "[no LDS reads pending]" not in line:
                bad.append(f"{os.path.basename(path)}:{i + 1}: {line.strip()[:100]}")
    assert seen >= 25, seen           # (the scan found the barriers it is meant to police)
    assert not bad, "raw s_barrier without an LDS-read wait or a reason:\n" + "\n".join(bad)
