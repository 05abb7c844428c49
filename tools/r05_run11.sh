#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run11
mkdir -p $O
export TMPDIR=/tmp
for l in hip hip_dbg r04; do python -c "import ctypes; ctypes.CDLL('miphei-vit_amd/libmiphei_$l.so')" || { echo "lib $l does not load"; exit 9; }; done
for r in 1 2 3; do
  for v in r04 hip_dbg; do
    echo "lib $v" >> $O/ab.txt
    MIPHEI_LIB=miphei-vit_amd/libmiphei_$v.so timeout 600 python tools/bench_lib.py --no-cpu-baseline --steps 30 --warmup 8 --probe 0 --comm-standin 0 2>> $O/ab.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $O/ab.txt
  done
done
cat $O/ab.txt
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch_test -- python3 $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --comm-standin 0 > $OLDPWD/$O/pmc_test.out 2> $OLDPWD/$O/pmc_test.err; echo "pmc rc $?" )
ls /tmp/pmc_fetch_test 2>/dev/null | head -3; tail -3 $O/pmc_test.err
