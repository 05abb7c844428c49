#!/bin/bash
# SQ counter breakdown of the head kernels inside the training step (separate rocprofv3 --pmc passes of bench.py, 2 steps)
R=$(pwd); O=$R/gpurun_out/hc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $O/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, re
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
want = ("gate_fwd", "conv_fwd_kernel", "conv_bwd_kernel", "gate_bwd_reduce", "gate_bwd_apply", "ln_bwd")
for f in glob.glob("gpurun_out/hc/p*/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").strip()
        if not any(w in k for w in want): continue
        a = acc[row["Counter_Name"]][k]; a[0] += 1; a[1] += float(row["Counter_Value"])
kern = sorted({k for c in acc.values() for k in c})
out = [f"{'counter (avg per dispatch)':30s}" + "".join(f"{k[:20]:>22s}" for k in kern)]
for c in sorted(acc): out.append(f"{c:30s}" + "".join(f"{acc[c][k][1] / max(1, acc[c][k][0]):22.4g}" for k in kern))
open("gpurun_out/hc/heads_counters.txt", "w").write("\n".join(out) + "\n"); print("\n".join(out))
PY
rm -rf $O/p*
