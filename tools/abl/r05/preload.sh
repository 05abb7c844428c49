#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/preload; mkdir -p $O; : > $O/log.txt
timeout 900 python -m pytest tests/test_gemm_ws_gpu.py tests/test_gemm_gpu.py -x -q 2>&1 | tail -1 >> $O/log.txt
WS_TIMING_ONLY=dproj,qkv,fc1+swiglu MIPHEI_LIB=miphei-vit_amd/csrc/variants/libmiphei_tm.so python tools/ws_timing.py 2>/dev/null | grep -v amdgpu | grep "warm\|first request" | cut -c1-420 >> $O/log.txt
bash tools/abl/step_ab.sh > /dev/null 2>&1
cut -c1-100 gpurun_out/step_ab/log.txt >> $O/log.txt
cat $O/log.txt
