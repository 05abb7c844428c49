# fc1 shapes on the product library vs the measurement library
echo "== product"; python3 tools/bench_vs_blas.py 2>/dev/null | grep "fc1\|sq8k"; python3 tools/bench_epi.py 2>/dev/null | grep fc1
echo "== dbg"; MIPHEI_DBG_LIB=1 python3 tools/bench_vs_blas.py 2>/dev/null | grep "fc1\|sq8k"; MIPHEI_DBG_LIB=1 python3 tools/bench_epi.py 2>/dev/null | grep fc1
for r in 1 2; do
echo "product"; python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | cut -c1-140
echo "dbg"; python3 tools/bench_dbg.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | cut -c1-140
done
