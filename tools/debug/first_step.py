"""Per training step of a rocprofv3 --kernel-trace csv run of bench.py (steps end at adam_kernel): number of launches, sum of kernel durations,
wall time, mean duration of the first / last ten store GEMMs.  Used to check that the 2-4 ms the FIRST timed step costs over the others (bench.py
`unprobed.step_ms`) is not kernel time: under the trace every step sums to the same 34.4-34.7 ms.   python tools/debug/first_step.py <trace dir>"""
import csv, glob, sys, re
f = glob.glob(f"{sys.argv[1]}/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
print("steps found", len(adam))
prev = 0
for k in range(len(adam)):
    lo = adam[k-1]+1 if k else 0
    step = rows[lo:adam[k]+1]
    dur = sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in step)/1e3
    wall = (int(step[-1]["End_Timestamp"])-int(step[0]["Start_Timestamp"]))/1e3
    # duration of the first 40 WS STORE gemms
    g=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in step if "gemm_ws_kernel<0" in r["Kernel_Name"]]
    print(k, len(step), "kernel-sum %.0f us wall %.0f us  first10 store-gemm avg %.1f  last10 avg %.1f" % (dur, wall, sum(g[:10])/10 if len(g)>=10 else 0, sum(g[-10:])/10 if len(g)>=10 else 0))
