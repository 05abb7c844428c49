# same-box A/B: product library vs measurement library (build the other arm first: make -C miphei-vit_amd/csrc dbg EXTRA=-D...),
# or dispatch knobs of the measurement library (MVIT_GEMM_* environment variables in front of tools/bench_dbg.py)
for r in 1 2 3; do
echo "product $(python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | cut -c60-110)"
echo "dbg     $(python3 tools/bench_dbg.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | cut -c60-110)"
done
