#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/kernarg; mkdir -p $O; : > $O/log.txt
for r in 1 2 3; do
echo "default          $(python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 --comm-standin 0 2>/dev/null | cut -c60-110)" >> $O/log.txt
echo "DEV_KERNARG=1    $(HIP_FORCE_DEV_KERNARG=1 python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 --comm-standin 0 2>/dev/null | cut -c60-110)" >> $O/log.txt
done
cat $O/log.txt
