cd "$(dirname "$0")/../.."
run() { echo "== $1"; env $2 python tools/gemm_power.py 2>&1 | grep -E "random ours"; }
run "8-wave 256x256 (default for 8192^3)" "X=1"
run "8-wave 256x128" "MVIT_GEMM_HUGE_MIN_TILES=100000000"
run "4-wave 256x128" "MVIT_GEMM_HUGE_MIN_TILES=100000000 MVIT_GEMM_W4=2"
run "4-wave 256x256" "MVIT_GEMM_W4=1"
