#!/bin/bash
# SQ counter breakdown of chosen kernels inside the training step: separate rocprofv3 --pmc passes of `bench.py --steps 2` (each pass
# <= 8 SQ counters, never combined with a trace domain), per-dispatch averages per kernel.
#   bash tools/sq_counters.sh <out.txt> <kernel-name substring> [...]        e.g.  gemm_kernel attn_ ln_fwd_lora
# Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_BUSY_CYCLES is
# summed over SEs/XCDs; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
R=$(cd "$(dirname "$0")/.." && pwd); OUT=$1; shift; WANT="$*"
O=$R/gpurun_out/sqc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM" \
         "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $O/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --probe 0 > /dev/null 2> $O/p$i.err
done
cd $R
python3 - "$OUT" $WANT <<'PY'
import csv, glob, re, sys
from collections import defaultdict
out_path, want = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/sqc/p*/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").replace("mvit_gemm::", "").strip()
        if not any(w in k for w in want): continue
        a = acc[row["Counter_Name"]][k]; a[0] += 1; a[1] += float(row["Counter_Value"])
kern = sorted({k for c in acc.values() for k in c}, key=lambda k: -acc.get("SQ_WAVE_CYCLES", {}).get(k, [0, 0])[1])
W = max(24, max((len(k) for k in kern), default=24) + 2)
out = [f"{'counter (avg per dispatch)':30s}" + "".join(f"{k:>{W}s}" for k in kern)]
for c in sorted(acc): out.append(f"{c:30s}" + "".join(f"{acc[c][k][1] / max(1, acc[c][k][0]):{W}.5g}" for k in kern))
def g(c, k): a = acc.get(c, {}).get(k); return a[1] / a[0] if a and a[0] else float("nan")
out.append("")
out.append("derived (per dispatch):")
rows = [("dispatches seen", lambda k: acc["SQ_WAVES"][k][0] if k in acc.get("SQ_WAVES", {}) else 0),
        ("shader clocks (GUI_ACTIVE/8)", lambda k: g("GRBM_GUI_ACTIVE", k) / 8),
        ("MFMA busy / (1024 SIMD x clk)", lambda k: g("SQ_VALU_MFMA_BUSY_CYCLES", k) / (1024 * g("GRBM_GUI_ACTIVE", k) / 8)),
        ("wave-cycles x4 / (waves x clk)", lambda k: 4 * g("SQ_WAVE_CYCLES", k) / (g("SQ_WAVES", k) * g("GRBM_GUI_ACTIVE", k) / 8)),
        ("WAIT_ANY / WAVE_CYCLES", lambda k: g("SQ_WAIT_ANY", k) / g("SQ_WAVE_CYCLES", k)),
        ("WAIT_INST_ANY / WAVE_CYCLES", lambda k: g("SQ_WAIT_INST_ANY", k) / g("SQ_WAVE_CYCLES", k)),
        ("WAIT_INST_LDS / WAVE_CYCLES", lambda k: g("SQ_WAIT_INST_LDS", k) / g("SQ_WAVE_CYCLES", k)),
        ("ACTIVE_INST_ANY / WAVE_CYCLES", lambda k: g("SQ_ACTIVE_INST_ANY", k) / g("SQ_WAVE_CYCLES", k)),
        ("VALU insts per MFMA", lambda k: g("SQ_INSTS_VALU", k) / g("SQ_INSTS_MFMA", k)),
        ("SALU insts per MFMA", lambda k: g("SQ_INSTS_SALU", k) / g("SQ_INSTS_MFMA", k)),
        ("LDS insts per MFMA", lambda k: g("SQ_INSTS_LDS", k) / g("SQ_INSTS_MFMA", k)),
        ("VMEM insts per MFMA", lambda k: g("SQ_INSTS_VMEM", k) / g("SQ_INSTS_MFMA", k)),
        ("LDS bank conflict / LDS active", lambda k: g("SQ_LDS_BANK_CONFLICT", k) / g("SQ_LDS_IDX_ACTIVE", k))]
for name, fn in rows:
    vals = []
    for k in kern:
        try: vals.append(f"{fn(k):{W}.4g}")
        except Exception: vals.append(f"{'-':>{W}s}")
    out.append(f"{name:30s}" + "".join(vals))
open(out_path, "w").write("\n".join(out) + "\n"); print("\n".join(out))
PY
rm -rf $O/p*/
