#!/bin/bash
# the per-class gradient-error ceilings of tests/test_full_size_gpu.py: the printed values over repeats (advisor: a max over tensors from one run)
cd $GRAFT_REPO_ROOT
O=gpurun_out/ceil_reps; mkdir -p $O; : > $O/log.txt
for r in 1 2 3; do
timeout 1500 python -m pytest tests/test_full_size_gpu.py -x -q -m gpu -s -k "forward_loss_gradnorm" 2>&1 | grep "autocast\|passed\|failed\|worst decoder" >> $O/log.txt
done
cat $O/log.txt
