"""Per-kernel MFMA-pipe utilisation and fabric (HBM-side) bandwidth of one training step, from separate rocprofv3 passes of
`bench.py --steps 2 --warmup 1 --no-cpu-baseline` (csv output):
    argv: <kernel-trace dir> <pmc dir with SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE> <pmc FETCH_SIZE dir> <pmc WRITE_SIZE dir>
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x shader clocks of the dispatch), shader clocks = GRBM_GUI_ACTIVE / 8
(the counter is summed over the 8 XCDs; MI355X_MICROARCH.md).  Bandwidth = (2*FETCH_SIZE + WRITE_SIZE) KB (gfx950 correction) /
kernel duration of the trace pass, against the 8 TB/s HBM3E peak (Infinity-Cache hits are included in the byte count).
"""
import csv, glob, re, sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"\(.*", "", n).replace("void ", "").strip()


def counters(d, names):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] in names:
            a = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return acc


trace = defaultdict(lambda: [0, 0.0])
f = glob.glob(f"{sys.argv[1]}/**/*_kernel_trace.csv", recursive=True)[0]
for row in csv.DictReader(open(f)):
    t = trace[short(row["Kernel_Name"])]
    t[0] += 1
    t[1] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3    # us
sq = counters(sys.argv[2], {"SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"})
fe = counters(sys.argv[3], {"FETCH_SIZE"})
wr = counters(sys.argv[4], {"WRITE_SIZE"})
total = sum(v[1] for k, v in trace.items())
print(f"{'kernel':58s} {'calls':>6s} {'avg_us':>8s} {'time %':>7s} {'MFMA util':>10s} {'GB/s':>8s} {'of 8 TB/s':>10s}")
for k, (n, us) in sorted(trace.items(), key=lambda kv: -kv[1][1])[:28]:
    if "at::native" in k or "rocclr" in k:
        continue
    mf = ""
    if k in sq and sq[k]["GRBM_GUI_ACTIVE"][1] > 0:
        busy, act = sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"][1], sq[k]["GRBM_GUI_ACTIVE"][1]
        mf = f"{busy / (1024.0 * act / 8.0):10.3f}"
    bw = frac = ""
    if k in fe and k in wr:
        kb = 2 * fe[k]["FETCH_SIZE"][1] / fe[k]["FETCH_SIZE"][0] + wr[k]["WRITE_SIZE"][1] / wr[k]["WRITE_SIZE"][0]
        gbs = kb * 1024 / (us / n * 1e-6) / 1e9
        bw, frac = f"{gbs:8.0f}", f"{gbs / 8000:10.3f}"
    print(f"{k[:58]:58s} {n:6d} {us / n:8.1f} {100 * us / total:7.2f} {mf:>10s} {bw:>8s} {frac:>10s}")
