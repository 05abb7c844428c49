"""Scan gfx950 assembly for a code shape suspected in round 4 (attention forward): a VALU instruction that rewrites a source register
(SrcA / SrcB) of an MFMA issued a few instructions earlier.  (NOT the cause in the end -- tools/probes/mfma_war.hip shows the hardware
interlocks it, and the wrong slabs were LDS reads in flight across s_barrier; kept as a scanner for the pattern.)
Usage: python tools/debug/scan_mfma_war.py file.s [window]"""
import re, sys

def regs(tok):
    tok = tok.strip()
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    if m:
        return {int(m.group(1))}
    return set()

path, window = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 6
kernel, lines = None, []
for raw in open(path):
    l = raw.strip()
    m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", l)
    if m and not l.startswith(".L"):
        kernel = m.group(1)
        continue
    if not l or l.startswith(";") or l.startswith("."):
        if l.startswith(".LBB"):
            lines.append((kernel, "LABEL", l))
        continue
    lines.append((kernel, "I", l.split(";")[0].strip()))
hits = {}
for i, (k, kind, l) in enumerate(lines):
    if kind != "I" or not l.startswith("v_mfma"):
        continue
    ops = l.split(None, 1)[1].split(",")
    src = regs(ops[1]) | regs(ops[2])
    n = 0
    for k2, kind2, l2 in lines[i + 1:]:
        if kind2 == "LABEL" or k2 != k:
            break
        if l2.startswith(("s_", "buffer_", "global_", "ds_", ";;")):
            continue            # (LDS / VMEM returns arrive tens of cycles later; scalar instructions do not write VGPRs)
        n += 1
        if n > window:
            break
        if l2.startswith("v_mfma"):
            continue
        if l2.startswith("v_") and not l2.startswith(("v_cmp", "v_cmpx")):
            dst = regs(l2.split(None, 1)[1].split(",")[0])
            if dst & src:
                hits.setdefault(k, []).append((n, l, l2))
                break
for k, v in hits.items():
    print(f"{k}: {len(v)} MFMA source rewrites within {window} vector instructions")
    for n, a, b in v[:4]:
        print(f"    +{n}: {a[:70]}  <-  {b[:60]}")
if not hits:
    print("no VALU rewrite of an MFMA source within", window, "vector instructions")
