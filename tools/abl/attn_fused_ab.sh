#!/bin/bash
# round 6: attention tests + same-box A/B of the one-pass backward (product library) against the two-kernel form (dbg library, MVIT_ATTN_FUSED=0)
cd $GRAFT_REPO_ROOT
# variant libraries do not travel (.gpurunignore): build the measurement library on the box when it is not there
[ -f miphei-vit_amd/csrc/variants/libmiphei_hip_dbg.so ] || make -C miphei-vit_amd/csrc -j16 dbg > /dev/null 2>&1
O=gpurun_out/attn_fused; mkdir -p $O; : > $O/log.txt
timeout 900 python -m pytest tests/test_attention_gpu.py -x -q 2>&1 | tail -15 >> $O/log.txt
for r in 1 2 3; do
  for N in 329 257; do
    echo "fused   $(python tools/bench_attn.py $N ours 2>/dev/null | grep N=)" >> $O/log.txt
    echo "2-kern  $(MIPHEI_DBG_LIB=1 MVIT_ATTN_FUSED=0 python tools/bench_attn.py $N ours 2>/dev/null | grep N=)" >> $O/log.txt
  done
done
cat $O/log.txt
