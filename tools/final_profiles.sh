# Round-end measurement pass on the GPU box: bench lines, rocprofv3 kernel stats and the two PMC passes -> gpurun_out/final/
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
python3 bench.py > $O/bench_train.json 2> $O/bench_train.err
python3 bench.py --metrics 1 --no-cpu-baseline > $O/bench_train_metrics_on.json 2>/dev/null
python3 bench.py --mode infer --batch 64 --steps 20 --warmup 5 > $O/bench_infer_b64.json 2>/dev/null
python3 bench.py --img 512 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train_512.json 2>/dev/null
python3 bench.py --mode embed --batch 64 --steps 20 --warmup 5 > $O/bench_embed.json 2>/dev/null
python3 bench.py --generator unet_lora --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_unetr_train.json 2>/dev/null
python3 bench.py --generator unet_lora --mode infer --batch 64 --steps 10 --warmup 3 > $O/bench_unetr_infer.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
db=$(ls $O/stats/*/*.db | head -1); python3 tools/prof_summary.py $db 60 > $O/kernel_stats.txt
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > $O/pmc_traffic.txt
tail -n 3 $O/*.json | cut -c1-300; head -20 $O/kernel_stats.txt; cat $O/pmc_traffic.txt
