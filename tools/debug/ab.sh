# same-box A/B: product library vs measurement library (build the other arm first: make -C miphei-vit_amd/csrc dbg EXTRA=-D...)
timeout 900 python3 -m pytest tests/test_generator_gpu.py tests/test_training_gpu.py tests/test_deterministic_gpu.py tests/test_heads_gpu.py -x -q 2>&1 | tail -2
for r in 1 2; do
echo "product"; python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | cut -c1-140
echo "dbg"; python3 tools/bench_dbg.py --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | cut -c1-140
done
export TMPDIR=/tmp
O=gpurun_out/seq; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
db=$(ls $O/stats/*/*.db 2>/dev/null | head -1)
python3 tools/prof_summary.py $db 90 | grep -i "bn_relu\|conv_bwd_dz\|upsample"
rm -rf $O/stats
