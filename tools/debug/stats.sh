export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_heads_gpu.py tests/test_deterministic_gpu.py tests/test_generator_gpu.py -x -q 2>&1 | tail -2
O=gpurun_out/seq; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
db=$(ls $O/stats/*/*.db 2>/dev/null | head -1)
python3 tools/prof_summary.py $db 70 > $O/kstats.txt
rm -rf $O/stats
grep -i "dw3\|gate_\|bn_from\|moments\|conv_fwd\|conv_bwd" $O/kstats.txt
