#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/fold; mkdir -p $O
python -c "import ctypes; [ctypes.CDLL('miphei-vit_amd/'+n) for n in ('libmiphei_hip.so','libmiphei_hip_dbg.so')]; print('libs load')" > $O/log.txt 2>&1
timeout 1500 python -m pytest tests/test_gemm_gpu.py tests/test_generator_gpu.py tests/test_full_size_gpu.py -x -q 2>&1 | tail -5 >> $O/log.txt
for r in 1 2; do
echo "fold 0: $(python bench.py --mode infer --batch 64 --steps 30 --warmup 5 --no-cpu-baseline --bn-fold 0 2>/dev/null | cut -c1-160)" >> $O/log.txt
echo "fold 1: $(python bench.py --mode infer --batch 64 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-160)" >> $O/log.txt
done
cat $O/log.txt
