#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run16
mkdir -p $O
export TMPDIR=/tmp
for l in hip hip_dbg; do python -c "import ctypes; ctypes.CDLL('miphei-vit_amd/libmiphei_$l.so')" || { echo "lib $l does not load"; exit 9; }; done
timeout 900 python -m pytest tests/test_gemm_ws_gpu.py tests/test_gemm_gpu.py -x -q > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
for r in 1 2 3; do
  for v in 2 1; do
    echo "MVIT_GEMM_WS_BAND=$v" >> $O/ab.txt
    MVIT_GEMM_WS_BAND=$v timeout 600 python tools/bench_dbg.py --no-cpu-baseline --steps 30 --warmup 8 --probe 0 --comm-standin 0 2>> $O/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O/ab.txt
  done
done
timeout 300 python tools/xcd_speed.py 2>&1 | grep -v amdgpu.ids > $O/xcd_speed.txt
timeout 1500 python -m pytest tests/test_training_gpu.py tests/test_generator_gpu.py tests/test_full_size_gpu.py -x -q > $O/pytest_model.log 2>&1
echo "pytest rc $?" >> $O/pytest_model.log
tail -3 $O/pytest.log; cat $O/ab.txt; cat $O/xcd_speed.txt; tail -3 $O/pytest_model.log
