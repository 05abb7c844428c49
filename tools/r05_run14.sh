#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run14
mkdir -p $O
export TMPDIR=/tmp
for l in hip hip_dbg tm dv3nm tm_dv3nm; do python -c "import ctypes; ctypes.CDLL('miphei-vit_amd/libmiphei_$l.so')" || { echo "lib $l does not load"; exit 9; }; done
timeout 600 python -m pytest tests/test_gemm_ws_gpu.py -x -q > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
for v in tm tm_dv3nm; do
  echo "== $v" >> $O/ws_timing.txt
  WS_TIMING_ONLY=dfc2+dswiglu MIPHEI_LIB=miphei-vit_amd/libmiphei_$v.so timeout 300 python tools/ws_timing.py >> $O/ws_timing.txt 2>&1
done
for r in 1 2; do
  for v in hip_dbg dv3nm; do
    echo "lib $v" >> $O/ab.txt
    MIPHEI_LIB=miphei-vit_amd/libmiphei_$v.so timeout 600 python tools/bench_lib.py --no-cpu-baseline --steps 30 --warmup 8 --probe 0 --comm-standin 0 2>> $O/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O/ab.txt
  done
done
tail -3 $O/pytest.log; grep -v amdgpu.ids $O/ws_timing.txt | grep "==\|warm" | cut -c1-300; cat $O/ab.txt
