cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
export MVIT_GEMM_W4=1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/l2a -- python3 tools/gemm_ablate.py - > /dev/null 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d gpurun_out/l2b -- python3 tools/gemm_ablate.py - > /dev/null 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d gpurun_out/l2c -- python3 tools/gemm_ablate.py - > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, re
from collections import defaultdict
for d in ("l2a","l2b","l2c"):
    fs = glob.glob(f"gpurun_out/{d}/**/*_counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(fs[0])):
        if "gemm_kernel" not in row["Kernel_Name"]: continue
        k = re.sub(r"\(.*", "", row["Kernel_Name"])[-40:] + " grid=" + row.get("Grid_Size","")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, c in acc.items():
        print(d, k, {n: (len(v), sum(v)/len(v)) for n, v in c.items()})
PY
