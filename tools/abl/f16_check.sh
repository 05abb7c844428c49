#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/f16_check; mkdir -p $O; : > $O/log.txt
timeout 1500 python -m pytest tests/test_generator_gpu.py tests/test_full_size_gpu.py -x -q -m gpu -k "half or registry or hipgraph or fold" -s -rA 2>&1 | grep "passed\|failed\|worst-channel\|PASSED\|FAILED" | tail -20 >> $O/log.txt
python3 bench.py --mode embed --batch 64 --steps 20 --warmup 5 2>/dev/null | cut -c1-200 >> $O/log.txt
python3 bench.py --mode infer --batch 64 --steps 20 --warmup 5 --dtype fp16 --probe 0 2>/dev/null | cut -c1-200 >> $O/log.txt
python3 bench.py --mode infer --batch 64 --steps 20 --warmup 5 --probe 0 2>/dev/null | cut -c1-200 >> $O/log.txt
cat $O/log.txt
