#!/bin/bash
# round 6: the GPU test files that exercise the attention backward inside the model + the headline bench, one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_check; mkdir -p $O; : > $O/log.txt
timeout 1500 python -m pytest tests/test_attention_gpu.py tests/test_deterministic_gpu.py tests/test_training_gpu.py tests/test_full_size_gpu.py tests/test_generator_gpu.py tests/test_unetr_golden.py -x -q -m gpu 2>&1 | tail -4 >> $O/log.txt
for r in 1 2; do
python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 --comm-standin 0 2>/dev/null | cut -c1-160 >> $O/log.txt
done
cat $O/log.txt
