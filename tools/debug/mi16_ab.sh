timeout 900 python3 -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -3
echo "== MI16 (product)"; python3 tools/bench_vs_blas.py 2>/dev/null; python3 tools/bench_epi.py 2>/dev/null
echo "== MI32 (dbg)"; MIPHEI_DBG_LIB=1 python3 tools/bench_vs_blas.py 2>/dev/null; MIPHEI_DBG_LIB=1 python3 tools/bench_epi.py 2>/dev/null
for r in 1 2; do
echo "MI16"; python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | cut -c1-140
echo "MI32"; python3 tools/bench_dbg.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | cut -c1-140
done
