#!/bin/bash
# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the three attention kernels: HEAD library vs the working tree
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/attn_lds; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MIPHEI_LIB=$R/miphei-vit_amd/csrc/variants/libmiphei_ab_head.so rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS --output-format csv -d $O/head -- python3 $R/tools/bench_attn.py 329 ours > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS --output-format csv -d $O/new -- python3 $R/tools/bench_attn.py 329 ours > /dev/null 2>&1
cd $R
python3 - <<'PY' > $O/summary.txt
import csv, glob, collections
for tag in ("head", "new"):
    f = glob.glob(f"gpurun_out/attn_lds/{tag}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "attn_" not in k: continue
        k = k.split("(")[0].split("::")[-1]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE": n[k] += 1
    for k in sorted(acc):
        a = acc[k]; c = max(1, n[k])
        print(f"{tag:5s} {k:24s} LDS_IDX_ACTIVE {a['SQ_LDS_IDX_ACTIVE']/c:12.0f}  BANK_CONFLICT {a['SQ_LDS_BANK_CONFLICT']/c:12.0f} ({100*a['SQ_LDS_BANK_CONFLICT']/max(1,a['SQ_LDS_IDX_ACTIVE']):4.1f} %)  ADDR_CONFLICT {a['SQ_LDS_ADDR_CONFLICT']/c:10.0f}")
PY
cat $O/summary.txt
rm -rf $O/head $O/new
