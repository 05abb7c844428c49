#!/bin/bash
# same-box, interleaved A/B of the training step: library A (MIPHEI_LIB path, default the saved HEAD build) vs the working tree's product library
# (library A is a build of an OLDER commit -- git worktree add /tmp/base <commit>; make -C /tmp/base/miphei-vit_amd/csrc LIB=<repo>/miphei-vit_amd/csrc/variants/libmiphei_base.so --
#  and variant libraries do not travel by default: take the csrc/variants/ line out of .gpurunignore for the call)
cd $GRAFT_REPO_ROOT
A=${1:-miphei-vit_amd/csrc/variants/libmiphei_base.so}
O=gpurun_out/step_ab; mkdir -p $O; : > $O/log.txt
python -c "import ctypes; [ctypes.CDLL(n) for n in ('miphei-vit_amd/libmiphei_hip.so','$A')]; print('libs load')" >> $O/log.txt 2>&1
for r in 1 2 3; do
echo "A   $(MIPHEI_LIB=$A python3 tools/bench_lib.py --no-cpu-baseline --steps 40 --warmup 8 --comm-standin 0 2>/dev/null | cut -c1-110)" >> $O/log.txt
echo "new $(python3 bench.py --no-cpu-baseline --steps 40 --warmup 8 --comm-standin 0 2>/dev/null | cut -c1-110)" >> $O/log.txt
done
cat $O/log.txt
