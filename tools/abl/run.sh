cd "$(dirname "$0")/../.."
for w4 in 0; do
  export MVIT_GEMM_W4=$w4
  if [ $w4 = 0 ]; then export MVIT_GEMM_HUGE_MIN_TILES=100000; else unset MVIT_GEMM_HUGE_MIN_TILES; fi
  echo "== W4=$w4"
  for l in miphei-vit_amd/libmiphei_hip.so tools/abl/*.so; do python tools/gemm_ablate.py $l 2>&1 | tail -2; done
done
python -m pytest tests/test_gemm_gpu.py -q 2>&1 | tail -2
