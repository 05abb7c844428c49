#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/wmap; mkdir -p $O; : > $O/log.txt
MIPHEI_LIB=miphei-vit_amd/csrc/variants/libmiphei_snk.so timeout 900 python -m pytest tests/test_gemm_ws_gpu.py -x -q 2>&1 | tail -2 >> $O/log.txt
bash tools/abl/step_ab.sh miphei-vit_amd/csrc/variants/libmiphei_snk.so > /dev/null 2>&1
cat gpurun_out/step_ab/log.txt | cut -c1-100 >> $O/log.txt
cat $O/log.txt
