cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pA -- python3 bench.py --steps 5 --warmup 2 > /dev/null 2>&1
export MVIT_GEMM_W4_MIN_TILES=100000
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pB -- python3 bench.py --steps 5 --warmup 2 > /dev/null 2>&1
for d in pA pB; do echo "== $d"; f=$(ls gpurun_out/$d/*/*kernel_stats.csv | head -1); head -14 $f | cut -d, -f1-4 | cut -c1-150; done
