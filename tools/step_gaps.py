"""GPU idle time inside one training step: python tools/step_gaps.py <rocprofv3 --kernel-trace csv dir>
Takes the launches between the last two adam_kernel launches as one step; prints busy time, wall time and the largest gaps."""
import csv, glob, sys
f = glob.glob(f"{sys.argv[1]}/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
a, b = adam[-2] + 1, adam[-1] + 1
step = rows[a:b]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e3
wall = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
gaps = []
for p, q in zip(step, step[1:]):
    g = (int(q["Start_Timestamp"]) - int(p["End_Timestamp"])) / 1e3
    gaps.append((g, p["Kernel_Name"][:50], q["Kernel_Name"][:50]))
print(f"launches {len(step)}  busy {busy:.1f} us  wall {wall:.1f} us  idle {wall - busy:.1f} us ({100 * (wall - busy) / wall:.1f} %)")
pos = [g for g in gaps if g[0] > 0]
print(f"positive gaps: {len(pos)}, mean {sum(g[0] for g in pos) / max(1, len(pos)):.2f} us")
for g in sorted(gaps, reverse=True)[:12]:
    print(f"{g[0]:8.1f} us  {g[1]} -> {g[2]}")
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
import re
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("mvit_gemm::", "")
    return n[:34]
for p, q in zip(step, step[1:]):
    g = (int(q["Start_Timestamp"]) - int(p["End_Timestamp"])) / 1e3
    a_ = agg[(short(p["Kernel_Name"]), short(q["Kernel_Name"]))]
    a_[0] += 1
    a_[1] += g
print("\nby transition (count, mean gap us, total us):")
for k, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{n:4d} {tot / n:7.2f} {tot:8.1f}   {k[0]} -> {k[1]}")
neg = sum(min(0.0, (int(q['Start_Timestamp']) - int(p['End_Timestamp'])) / 1e3) for p, q in zip(step, step[1:]))
print(f"sum of negative gaps (overlap): {neg:.1f} us")
