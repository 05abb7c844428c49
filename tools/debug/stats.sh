export TMPDIR=/tmp
O=gpurun_out/seq; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
db=$(ls $O/stats/*/*.db 2>/dev/null | head -1)
python3 tools/prof_summary.py $db 60 > $O/kstats.txt
rm -rf $O/stats
head -60 $O/kstats.txt
