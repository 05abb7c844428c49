#!/bin/bash
# measurement: the WS GEMM's K step on v_mfma_f32_32x32x16_bf16 (libmiphei_${VAR:-m32}.so, results wrong) against the 16x16x32 product form (libmiphei_tm.so)
cd $GRAFT_REPO_ROOT
O=gpurun_out/mfma32; mkdir -p $O; : > $O/log.txt
python -c "import ctypes; [ctypes.CDLL('miphei-vit_amd/csrc/variants/'+n) for n in ('libmiphei_tm.so','libmiphei_${VAR:-m32}.so')]; print('libs load')" >> $O/log.txt 2>&1
for rep in 1 2; do
for L in tm ${VAR:-m32}; do
  echo "== $L" >> $O/log.txt
  WS_TIMING_ONLY=dproj,dqkv,dfc1,qkv MIPHEI_LIB=miphei-vit_amd/csrc/variants/libmiphei_$L.so python tools/ws_timing.py 2>/dev/null | grep -v amdgpu.ids | grep "warm" | cut -c1-330 >> $O/log.txt
done
done
cat $O/log.txt
