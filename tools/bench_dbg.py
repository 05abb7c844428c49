"""bench.py on the measurement library (libmiphei_hip_dbg.so, `make -C miphei-vit_amd/csrc dbg`): same-box A/B of the dispatch knobs,
e.g.  MVIT_GEMM_HUGE_MIN_TILES=600 python tools/bench_dbg.py --no-cpu-baseline --steps 30"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miphei_vit_amd import _lib
_lib.LIB_PATH = _lib.DBG_LIB_PATH
import bench
sys.exit(bench.main())
