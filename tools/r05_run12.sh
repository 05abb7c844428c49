#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run12
mkdir -p $O
export TMPDIR=/tmp
for l in hip hip_dbg tm; do python -c "import ctypes; ctypes.CDLL('miphei-vit_amd/libmiphei_$l.so')" || { echo "lib $l does not load"; exit 9; }; done
timeout 900 python -m pytest tests/test_gemm_ws_gpu.py tests/test_gemm_gpu.py -x -q > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
for v in 1 0; do
  echo "== MVIT_GEMM_WS_PACKST=$v" >> $O/ws_timing.txt
  MVIT_GEMM_WS_PACKST=$v WS_TIMING_ONLY=dproj,qkv,dfc1 MIPHEI_LIB=miphei-vit_amd/libmiphei_tm.so timeout 300 python tools/ws_timing.py >> $O/ws_timing.txt 2>&1
done
for r in 1 2; do
  for v in 0 1; do
    echo "MVIT_GEMM_WS_PACKST=$v" >> $O/ab.txt
    MVIT_GEMM_WS_PACKST=$v timeout 600 python tools/bench_dbg.py --no-cpu-baseline --steps 30 --warmup 8 --probe 0 --comm-standin 0 2>> $O/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O/ab.txt
  done
done
timeout 1500 python -m pytest tests/test_training_gpu.py tests/test_generator_gpu.py tests/test_deterministic_gpu.py -x -q > $O/pytest_model.log 2>&1
echo "pytest rc $?" >> $O/pytest_model.log
tail -3 $O/pytest.log; grep -v amdgpu.ids $O/ws_timing.txt | grep "==\|warm" | cut -c1-300; cat $O/ab.txt; tail -3 $O/pytest_model.log
