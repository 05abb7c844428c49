#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/ho; mkdir -p $O; : > $O/log.txt
for r in 1 2 3; do
for L in hip ho2 ho10; do
echo "$L $(MIPHEI_LIB=miphei-vit_amd/csrc/variants/libmiphei_$L.so python3 tools/bench_lib.py --no-cpu-baseline --steps 40 --warmup 8 --comm-standin 0 2>/dev/null | cut -c60-100)" >> $O/log.txt
done
done
cat $O/log.txt
