"""Durations of every launch of one kernel (substring match) inside the last step of a `rocprofv3 --kernel-trace --output-format csv`
run of bench.py:  python tools/kernel_calls.py <trace dir> <kernel substring> [launches per step]"""
import csv, glob, sys
f = glob.glob(f"{sys.argv[1]}/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
for r in rows[-n:]:
    print(f'{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} us  grid {r.get("Grid_Size_X", "?")}')
