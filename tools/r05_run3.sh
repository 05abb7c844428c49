#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run3
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_ws_gpu.py tests/test_gemm_gpu.py -x -q > $O/pytest_gemm.log 2>&1
echo "pytest rc $?" >> $O/pytest_gemm.log
WS_TIMING_DUMP=$O/ws MIPHEI_LIB=miphei-vit_amd/libmiphei_tm.so timeout 300 python tools/ws_timing.py > $O/ws_timing.txt 2>&1
MVIT_GEMM_WS_RSINGLE=0 WS_TIMING_DUMP=$O/ws0 MIPHEI_LIB=miphei-vit_amd/libmiphei_tm.so timeout 300 python tools/ws_timing.py > $O/ws_timing_rsingle0.txt 2>&1
for r in 1 2; do
  for v in 0 1; do
    echo "MVIT_GEMM_WS_RSINGLE=$v" >> $O/ab.txt
    MVIT_GEMM_WS_RSINGLE=$v timeout 600 python tools/bench_dbg.py --no-cpu-baseline --steps 30 --warmup 8 --probe 0 --comm-standin 0 2>> $O/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O/ab.txt
  done
done
timeout 900 python -m pytest tests/test_training_gpu.py tests/test_generator_gpu.py tests/test_deterministic_gpu.py -x -q > $O/pytest_model.log 2>&1
echo "pytest rc $?" >> $O/pytest_model.log
tail -5 $O/pytest_gemm.log; tail -5 $O/pytest_model.log
cat $O/ws_timing.txt; echo ---- RSINGLE=0; grep -A1 "res" $O/ws_timing_rsingle0.txt; cat $O/ab.txt
