#!/bin/bash
# A/B of the forward attention kernel: HEAD library vs the working tree (product and dbg library with the rotation modes)
cd $GRAFT_REPO_ROOT
O=gpurun_out/attn_ab; mkdir -p $O
python -c "import ctypes; [ctypes.CDLL('miphei-vit_amd/'+n) for n in ('libmiphei_hip.so','libmiphei_hip_dbg.so','libmiphei_ab_head.so')]; print('libs load')" > $O/log.txt 2>&1
timeout 900 python -m pytest tests/test_attention_gpu.py -x -q >> $O/log.txt 2>&1
for rep in 1 2; do
for N in 329 1301; do
  echo "head N=$N" >> $O/log.txt; MIPHEI_LIB=miphei-vit_amd/csrc/variants/libmiphei_ab_head.so python tools/bench_attn.py $N ours >> $O/log.txt 2>&1
  echo "new  N=$N" >> $O/log.txt; python tools/bench_attn.py $N ours >> $O/log.txt 2>&1
done
for R in 0 1 2 3; do
  echo "dbg rot=$R N=329" >> $O/log.txt; MVIT_ATTN_ROT=$R MIPHEI_DBG_LIB=1 python tools/bench_attn.py 329 ours >> $O/log.txt 2>&1
done
done
tail -40 $O/log.txt
