"""pytest on a chosen build of the library: e.g. the dbg library with a dispatch knob set in the environment.
    MVIT_GEMM_WS4=1 python tools/pytest_lib.py miphei-vit_amd/csrc/variants/libmiphei_hip_dbg.so tests/test_gemm_ws_gpu.py -x -q
(shorthand for pytest --variant-lib PATH ...; tests/conftest.py takes the library from that option only, never from the environment)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest
sys.exit(pytest.main(["--variant-lib", sys.argv[1]] + sys.argv[2:]))
