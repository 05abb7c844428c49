#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/attn_fused_var; mkdir -p $O; : > $O/log.txt
for v in "$@"; do
  echo "=== $v" >> $O/log.txt
  MIPHEI_LIB=miphei-vit_amd/csrc/variants/libmiphei_$v.so python tools/debug/attn_fused_timing.py 2>&1 | grep -v "amdgpu.ids\|XCD" >> $O/log.txt
  echo "$v: $(MIPHEI_LIB=miphei-vit_amd/csrc/variants/libmiphei_$v.so python tools/bench_attn.py 329 ours 2>/dev/null | grep N=)" >> $O/log.txt
done
cat $O/log.txt
