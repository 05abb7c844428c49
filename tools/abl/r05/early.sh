#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/early; mkdir -p $O; : > $O/log.txt
python -c "import ctypes; [ctypes.CDLL('miphei-vit_amd/'+n) for n in ('libmiphei_hip.so','libmiphei_tm.so','libmiphei_hip_dbg.so','libmiphei_ab_head.so')]; print('libs load')" >> $O/log.txt 2>&1
timeout 900 python -m pytest tests/test_gemm_ws_gpu.py tests/test_gemm_gpu.py -x -q 2>&1 | tail -1 >> $O/log.txt
MIPHEI_LIB=miphei-vit_amd/csrc/variants/libmiphei_tm.so python tools/ws_timing.py 2>/dev/null | grep -v amdgpu | grep "warm\|first request" | awk '/warm/{l=$0} /first request/{n++; if(n%2==0) print substr(l,1,200) " | first req " $(NF-3)}' >> $O/log.txt
bash tools/abl/step_ab.sh > /dev/null 2>&1
cut -c1-100 gpurun_out/step_ab/log.txt >> $O/log.txt
cat $O/log.txt
