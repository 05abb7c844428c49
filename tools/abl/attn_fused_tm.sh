#!/bin/bash
cd $GRAFT_REPO_ROOT
# variant libraries do not travel (.gpurunignore): build the timing variant on the box when it is not there
[ -f miphei-vit_amd/csrc/variants/libmiphei_tm.so ] || make -C miphei-vit_amd/csrc -j16 BUILD=build_tm LIB=variants/libmiphei_tm.so EXTRA=-DMVIT_ATTN_TIMING > /dev/null 2>&1
O=gpurun_out/attn_fused_tm; mkdir -p $O
MIPHEI_LIB=miphei-vit_amd/csrc/variants/libmiphei_tm.so python tools/debug/attn_fused_timing.py 2>&1 | grep -v amdgpu.ids > $O/log.txt
timeout 600 python -m pytest tests/test_attention_gpu.py -x -q 2>&1 | tail -2 >> $O/log.txt
echo "product: $(python tools/bench_attn.py 329 ours 2>/dev/null | grep N=)" >> $O/log.txt
cat $O/log.txt
