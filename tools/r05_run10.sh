#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run10
mkdir -p $O
export TMPDIR=/tmp
for l in hip hip_dbg pl; do python -c "import ctypes; ctypes.CDLL('miphei-vit_amd/libmiphei_$l.so')" || { echo "lib $l does not load"; exit 9; }; done
for r in 1 2; do
  for v in hip_dbg pl; do
    echo "lib $v" >> $O/ab.txt
    MIPHEI_LIB=miphei-vit_amd/libmiphei_$v.so timeout 600 python tools/bench_lib.py --no-cpu-baseline --steps 30 --warmup 8 --probe 0 --comm-standin 0 2>> $O/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O/ab.txt
  done
done
cat $O/ab.txt; tail -3 $O/ab.err
