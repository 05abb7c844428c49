#!/bin/bash
# attention micro-benchmark: product library vs a compile-time variant (miphei-vit_amd/csrc/variants/libmiphei_snk.so)
cd $GRAFT_REPO_ROOT
O=gpurun_out/attn_var; mkdir -p $O; : > $O/log.txt
MIPHEI_LIB=miphei-vit_amd/csrc/variants/libmiphei_snk.so timeout 600 python -m pytest tests/test_attention_gpu.py -x -q 2>&1 | tail -1 >> $O/log.txt
for rep in 1 2 3; do
for N in 329 1301; do
echo "product $(python tools/bench_attn.py $N ours 2>/dev/null | grep N=)" >> $O/log.txt
echo "variant $(MIPHEI_LIB=miphei-vit_amd/csrc/variants/libmiphei_snk.so python tools/bench_attn.py $N ours 2>/dev/null | grep N=)" >> $O/log.txt
done
done
cat $O/log.txt
