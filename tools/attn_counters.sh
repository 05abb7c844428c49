#!/bin/bash
# SQ counter breakdown of the three attention kernels at the training shape (separate rocprofv3 --pmc passes of tools/bench_attn.py).
#   bash tools/attn_counters.sh <tag>     -> gpurun_out/<tag>/attn_counters.txt
set -u
T=${1:-attn}
R=$(pwd)
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS" \
         "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_CVT SQ_INSTS_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $O/p$i -- python3 $R/tools/bench_attn.py 329 ours > /dev/null 2>&1
done
cd $R
python3 tools/attn_counters.py $O > $O/attn_counters.txt
cat $O/attn_counters.txt
