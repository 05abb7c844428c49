#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run4
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_ws_gpu.py tests/test_attention_gpu.py -x -q > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
for v in tm tm_r2 tm_r8; do
  echo "== $v" >> $O/ws_timing_res.txt
  WS_TIMING_ONLY=proj+res,fc2+res MIPHEI_LIB=miphei-vit_amd/libmiphei_$v.so timeout 300 python tools/ws_timing.py >> $O/ws_timing_res.txt 2>&1
done
WS_TIMING_ONLY=fc1+swiglu,dfc2+dswiglu MIPHEI_LIB=miphei-vit_amd/libmiphei_tm.so timeout 300 python tools/ws_timing.py > $O/ws_timing_swiglu.txt 2>&1
timeout 300 python tools/bench_attn.py > $O/attn.txt 2>&1
for r in 1 2; do
  for v in 0 1; do
    echo "MVIT_GEMM_WS_RSINGLE=$v" >> $O/ab.txt
    MVIT_GEMM_WS_RSINGLE=$v timeout 600 python tools/bench_dbg.py --no-cpu-baseline --steps 30 --warmup 8 --probe 0 --comm-standin 0 2>> $O/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O/ab.txt
  done
done
tail -3 $O/pytest.log
grep -v amdgpu.ids $O/ws_timing_res.txt | cut -c1-330; grep -v amdgpu.ids $O/ws_timing_swiglu.txt | cut -c1-330; cat $O/attn.txt | tail -5; cat $O/ab.txt
