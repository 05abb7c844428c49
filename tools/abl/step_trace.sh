#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/step_trace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --probe 0 > /dev/null 2>&1
python3 tools/step_gaps.py $O/trace > $O/gaps.txt 2>&1
python3 tools/step_sequence.py $O/trace > $O/sequence.txt 2>&1
rm -rf $O/trace
cat $O/gaps.txt; head -14 $O/sequence.txt | cut -c1-150
