#!/bin/bash
# the bench's parity leg (HIP step vs fp32 CPU oracle at B = 16) a few times, working tree vs a saved library
cd $GRAFT_REPO_ROOT
O=gpurun_out/parity_reps; mkdir -p $O; : > $O/log.txt
for r in 1 2 3; do
for L in miphei-vit_amd/libmiphei_hip.so miphei-vit_amd/csrc/variants/libmiphei_ab_head.so; do
  MIPHEI_LIB=$L python3 tools/bench_lib.py --steps ${STEPS:-50} --warmup 10 --comm-standin 0 --cpu-repeats 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); p=d['parity']
print('$L', d['value'], 'relmse %.2e loss_err %.2e gnorm_err %.4f loss %.3f' % (p['worst_channel_rel_mse'], p['loss_rel_err'], p['grad_norm_rel_err'], p['loss']))" >> $O/log.txt 2>&1
done
done
cat $O/log.txt
