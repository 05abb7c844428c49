#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/attn_tm; mkdir -p $O
python -c "import ctypes; [ctypes.CDLL('miphei-vit_amd/'+n) for n in ('libmiphei_hip.so','libmiphei_tm.so','libmiphei_ab_head.so')]; print('libs load')" > $O/log.txt 2>&1
MIPHEI_LIB=miphei-vit_amd/csrc/variants/libmiphei_tm.so python tools/debug/attn_timing.py 2>&1 | grep -v "XCD\|amdgpu.ids" >> $O/log.txt
for rep in 1 2 3; do
echo head >> $O/log.txt; MIPHEI_LIB=miphei-vit_amd/csrc/variants/libmiphei_ab_head.so python tools/bench_attn.py 329 ours >> $O/log.txt 2>&1
echo new >> $O/log.txt; python tools/bench_attn.py 329 ours >> $O/log.txt 2>&1
done
timeout 600 python -m pytest tests/test_attention_gpu.py -x -q 2>&1 | tail -2 >> $O/log.txt
grep -v amdgpu.ids $O/log.txt
