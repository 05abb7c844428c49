#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/infer_ws; mkdir -p $O; : > $O/log.txt
for r in 1 2; do
for W in 15 0 13 14; do
echo "WS=$W $(MVIT_GEMM_WS=$W python3 tools/bench_dbg.py --mode infer --batch 64 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c35-75)" >> $O/log.txt
done
done
cat $O/log.txt
