export TMPDIR=/tmp
O=gpurun_out/seq; rm -rf $O; mkdir -p $O
python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 | cut -c1-200
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 tools/step_sequence.py $O/trace > $O/seq.txt
rm -rf $O/trace
head -20 $O/seq.txt | cut -c1-150
awk '{ if ($6 > 1.0) print }' $O/seq.txt | cut -c1-150 | head -40
