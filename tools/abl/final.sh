#!/bin/bash
# GPU test suite + smoke + the whole measurement pass, one box
cd $GRAFT_REPO_ROOT   # (BUILD_COMMIT -- git rev-parse HEAD > BUILD_COMMIT before the gpurun call -- stamps profiles/*_pmc_traffic.json)
O=gpurun_out/final; mkdir -p $O
python -c "import ctypes; ctypes.CDLL('miphei-vit_amd/libmiphei_hip.so'); print('lib loads')" > $O/log.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 >> $O/log.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O/log.txt 2>&1
bash tools/bench_all.sh r06 > $O/bench_all.txt 2>&1
tail -5 $O/log.txt; tail -60 $O/bench_all.txt
