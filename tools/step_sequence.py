"""Launch sequence of ONE training step (between the last two adam_kernel launches of a rocprofv3 --kernel-trace csv run), in
stream order: index, start offset, duration, gap to the previous launch, grid, kernel.  Attributes decoder / encoder time per
layer (the sequence is deterministic).   python tools/step_sequence.py <trace dir> [first] [last]"""
import csv, glob, re, sys
f = glob.glob(f"{sys.argv[1]}/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
step = rows[adam[-2] + 1:adam[-1] + 1]
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(step)
t0 = int(step[0]["Start_Timestamp"])
prev_end = t0
for i, r in enumerate(step):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if lo <= i < hi:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n)
        grid = "x".join(str(r.get(k, "?")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        wg = r.get("Workgroup_Size_X", "?")
        print(f"{i:4d} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  grid {grid:>14s} wg {wg:>4s}  {n[:100]}")
    prev_end = e
print(f"{len(step)} launches, {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
