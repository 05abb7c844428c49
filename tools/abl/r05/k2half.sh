#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/k2half; mkdir -p $O; : > $O/log.txt
python -c "import ctypes; [ctypes.CDLL('miphei-vit_amd/'+n) for n in ('libmiphei_hip.so','libmiphei_hip_dbg.so','libmiphei_ab_head.so')]; print('libs load')" >> $O/log.txt 2>&1
timeout 900 python -m pytest tests/test_gemm_ws_gpu.py tests/test_gemm_gpu.py tests/test_full_size_gpu.py -x -q 2>&1 | tail -2 >> $O/log.txt
bash tools/abl/step_ab.sh > /dev/null 2>&1
cat gpurun_out/step_ab/log.txt >> $O/log.txt
cat $O/log.txt
