O=gpurun_out/r03
mkdir -p $O
python3 bench.py > $O/bench_train.json 2> $O/bench_train.err
python3 bench.py --metrics 1 --no-cpu-baseline > $O/bench_train_metrics_on.json 2>/dev/null
python3 bench.py --img 512 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train_512_b4.json 2>/dev/null
MIPHEI_FORCE_DDP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_train_rccl_1rank.json 2>/dev/null
python3 bench.py --generator unet_lora --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_unetr_train.json 2>/dev/null
cut -c1-300 $O/bench_train.json
