#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/pprio; mkdir -p $O; : > $O/log.txt
timeout 900 python -m pytest tests/test_gemm_ws_gpu.py -x -q 2>&1 | tail -1 >> $O/log.txt
WS_TIMING_ONLY=dproj,qkv MIPHEI_LIB=miphei-vit_amd/csrc/variants/libmiphei_tm.so python tools/ws_timing.py 2>/dev/null | grep -v amdgpu | grep "warm\|first request" | cut -c1-420 >> $O/log.txt
for r in 1 2 3; do
echo "head   $(MIPHEI_LIB=miphei-vit_amd/csrc/variants/libmiphei_ab_head.so python3 tools/bench_lib.py --no-cpu-baseline --steps 40 --warmup 8 --comm-standin 0 2>/dev/null | cut -c60-100)" >> $O/log.txt
echo "prio1  $(python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 --comm-standin 0 2>/dev/null | cut -c60-100)" >> $O/log.txt
echo "prio2  $(MIPHEI_LIB=miphei-vit_amd/csrc/variants/libmiphei_snk.so python3 tools/bench_lib.py --no-cpu-baseline --steps 40 --warmup 8 --comm-standin 0 2>/dev/null | cut -c60-100)" >> $O/log.txt
done
cat $O/log.txt
