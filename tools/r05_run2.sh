#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run2
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_full_size_gpu.py -x -q -s -k "forward_loss_gradnorm or hipgraph_replay" > $O/pytest_fullsize.log 2>&1
echo "pytest rc $?" >> $O/pytest_fullsize.log
timeout 900 python -m pytest tests/test_ddp_two_ranks_gpu.py tests/test_gemm_gpu.py tests/test_gemm_ws_gpu.py -x -q > $O/pytest_ddp_gemm.log 2>&1
echo "pytest rc $?" >> $O/pytest_ddp_gemm.log
WS_TIMING_DUMP=$O/ws MIPHEI_LIB=miphei-vit_amd/libmiphei_tm.so timeout 300 python tools/ws_timing.py > $O/ws_timing.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --steps 30 --warmup 8 > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 8 --comm-standin 16,200 --probe 0 > $O/bench_standin200.json 2>> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 8 --comm-standin 32,200 --probe 0 > $O/bench_standin32.json 2>> $O/bench.err
grep -n "worst absolute" -A 5 $O/pytest_fullsize.log | head -40; tail -5 $O/pytest_fullsize.log
tail -5 $O/pytest_ddp_gemm.log
cat $O/ws_timing.txt
python - <<'PY'
import json
for f in ("bench","bench_standin200","bench_standin32"):
    try:
        d=json.loads(open(f"gpurun_out/r05_run2/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d.get("comm_overlap_probe"))
    except Exception as e: print(f, "ERR", e)
PY
tail -5 $O/bench.err
