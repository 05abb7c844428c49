"""Kernel launches of ONE training step (between the last two adam_kernel launches of a rocprofv3 --kernel-trace csv run), grouped by
name: count and total time.  python tools/step_kernels.py <trace dir> [substring filter]"""
import csv, glob, re, sys
from collections import defaultdict
f = glob.glob(f"{sys.argv[1]}/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
step = rows[adam[-2] + 1:adam[-1] + 1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = defaultdict(lambda: [0, 0.0])
for r in step:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    if flt in n:
        a = agg[n[:110]]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"{sum(v[0] for v in agg.values())} launches, {tot:.1f} us")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:4d} {t:9.1f} us  {k}")
