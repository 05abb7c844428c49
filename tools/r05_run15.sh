#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_run15
mkdir -p $O
export TMPDIR=/tmp
for l in hip hip_dbg r04; do python -c "import ctypes; ctypes.CDLL('miphei-vit_amd/libmiphei_$l.so')" || { echo "lib $l does not load"; exit 9; }; done
timeout 900 python -m pytest tests/test_conv_direct_gpu.py -x -q > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
for v in r04 hip r04 hip; do
  echo "== lib $v" >> $O/convs.txt
  ONLY=fus0,fus1,fus2,fus3 MIPHEI_LIB=miphei-vit_amd/libmiphei_$v.so timeout 300 python tools/bench_decoder_convs.py 2>&1 | grep -v amdgpu.ids | grep "fus" >> $O/convs.txt
done
tail -3 $O/pytest.log; cut -c1-60,150-400 $O/convs.txt
