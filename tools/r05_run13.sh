#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run13
mkdir -p $O
export TMPDIR=/tmp
for l in hip hip_dbg; do python -c "import ctypes; ctypes.CDLL('miphei-vit_amd/libmiphei_$l.so')" || { echo "lib $l does not load"; exit 9; }; done
MIPHEI_LIB=miphei-vit_amd/libmiphei_hip_dbg.so MVIT_GEMM_WS4=1 timeout 900 python tools/pytest_lib.py tests/test_gemm_ws_gpu.py tests/test_gemm_gpu.py -x -q > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
for v in 0 1; do
  echo "== MVIT_GEMM_WS4=$v" >> $O/vs.txt
  MVIT_GEMM_WS4=$v MIPHEI_LIB=miphei-vit_amd/libmiphei_hip_dbg.so timeout 300 python tools/bench_ws_abl.py >> $O/vs.txt 2>&1
done
for r in 1 2; do
  for v in 0 1 2 4; do
    echo "MVIT_GEMM_WS4=$v" >> $O/ab.txt
    MVIT_GEMM_WS4=$v timeout 600 python tools/bench_dbg.py --no-cpu-baseline --steps 30 --warmup 8 --probe 0 --comm-standin 0 2>> $O/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O/ab.txt
  done
done
tail -3 $O/pytest.log; grep -v amdgpu.ids $O/vs.txt; cat $O/ab.txt
