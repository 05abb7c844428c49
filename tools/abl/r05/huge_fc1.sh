#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/huge; mkdir -p $O; : > $O/log.txt
for r in 1 2 3; do
echo "ws   $(python3 tools/bench_dbg.py --no-cpu-baseline --steps 40 --warmup 8 --comm-standin 0 2>/dev/null | cut -c60-110)" >> $O/log.txt
echo "huge $(MVIT_GEMM_WS=13 python3 tools/bench_dbg.py --no-cpu-baseline --steps 40 --warmup 8 --comm-standin 0 2>/dev/null | cut -c60-110)" >> $O/log.txt
done
cat $O/log.txt
