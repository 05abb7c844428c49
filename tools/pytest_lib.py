"""pytest on a chosen build of the library (MIPHEI_LIB=path): e.g. the dbg library with a dispatch knob set in the environment.
    MIPHEI_LIB=miphei-vit_amd/libmiphei_hip_dbg.so MVIT_GEMM_WS4=1 python tools/pytest_lib.py tests/test_gemm_ws_gpu.py -x -q"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miphei_vit_amd import _lib
if os.environ.get("MIPHEI_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIPHEI_LIB"])
import pytest
sys.exit(pytest.main(sys.argv[1:]))
