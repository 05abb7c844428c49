# same-box A/B: product library vs measurement library (build the other arm first: make -C miphei-vit_amd/csrc dbg EXTRA=-D...)
timeout 900 python3 -m pytest tests/test_attention_gpu.py -x -q 2>&1 | tail -2
echo product; python3 tools/bench_attn.py 329 ours 2>/dev/null; python3 tools/bench_attn.py 1301 ours 2>/dev/null
echo dbg; MIPHEI_DBG_LIB=1 python3 tools/bench_attn.py 329 ours 2>/dev/null; MIPHEI_DBG_LIB=1 python3 tools/bench_attn.py 1301 ours 2>/dev/null
for r in 1 2 3; do
echo "product $(python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | cut -c60-110)"
echo "dbg     $(python3 tools/bench_dbg.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | cut -c60-110)"
done
