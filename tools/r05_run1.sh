#!/bin/bash
# round-5 GPU run 1: new parity tests, WS launch timing, power comparison, DMA-only tile ablation, baseline bench
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run1
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_full_size_gpu.py -x -q -s -k "forward_loss_gradnorm or hipgraph_replay" > $O/pytest_fullsize.log 2>&1
echo "pytest rc $?" >> $O/pytest_fullsize.log
MIPHEI_LIB=miphei-vit_amd/libmiphei_tm.so timeout 300 python tools/ws_timing.py > $O/ws_timing.txt 2>&1
timeout 300 python tools/gemm_power.py > $O/gemm_power.txt 2>&1
for t in 1 1000000; do
  echo "HUGE_MIN_TILES=$t (1 = 256x256 tile, 1000000 = 256x128 tile), non-WS kernel, DMA only (MVIT_ABLATE=6)" >> $O/dma_only_tiles.txt
  MVIT_GEMM_WS=0 MVIT_GEMM_HUGE_MIN_TILES=$t timeout 120 python tools/gemm_ablate.py miphei-vit_amd/libmiphei_abl6.so >> $O/dma_only_tiles.txt 2>&1
  echo "same, complete kernel (dbg library)" >> $O/dma_only_tiles.txt
  MVIT_GEMM_WS=0 MVIT_GEMM_HUGE_MIN_TILES=$t timeout 120 python tools/gemm_ablate.py miphei-vit_amd/libmiphei_hip_dbg.so >> $O/dma_only_tiles.txt 2>&1
done
echo "WS kernel, complete (dbg library)" >> $O/dma_only_tiles.txt
timeout 120 python tools/gemm_ablate.py miphei-vit_amd/libmiphei_hip_dbg.so >> $O/dma_only_tiles.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --steps 30 --warmup 8 > $O/bench_base.json 2> $O/bench_base.err
tail -c 3000 $O/pytest_fullsize.log
cat $O/ws_timing.txt $O/gemm_power.txt $O/dma_only_tiles.txt
cut -c1-400 $O/bench_base.json
