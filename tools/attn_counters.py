"""Per-dispatch averages of the SQ counters collected by tools/attn_counters.sh, one column per attention kernel."""
import csv, glob, re, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(f"{sys.argv[1]}/p*/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").strip()
        if "attn" not in k:
            continue
        a = acc[row["Counter_Name"]][k]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
kern = sorted({k for c in acc.values() for k in c})
print(f"{'counter (avg per dispatch)':34s}" + "".join(f"{k[:22]:>24s}" for k in kern))
for c in sorted(acc):
    print(f"{c:34s}" + "".join(f"{acc[c][k][1] / max(1, acc[c][k][0]):24.4g}" for k in kern))
