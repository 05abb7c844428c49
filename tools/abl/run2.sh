cd "$(dirname "$0")/../.."
echo "== default"; python tools/bench_vs_blas.py 2>&1 | grep "ours"
echo "== W4=2 (4-wave 256x128 wherever dense N%128==0)"; MVIT_GEMM_W4=2 MVIT_GEMM_HUGE_MIN_TILES=100000 python tools/bench_vs_blas.py 2>&1 | grep ours | sed 's/| hipBLASLt.*//'
echo "== W4=1 (4-wave 256x256 for huge)"; MVIT_GEMM_W4=1 python tools/bench_vs_blas.py 2>&1 | grep -E "fc1|sq8k" | sed 's/| hipBLASLt.*//'
