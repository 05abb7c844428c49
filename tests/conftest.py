import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

if os.environ.get("MIPHEI_DBG_LIB") == "1":
    # measurement runs only: the same tests against libmiphei_hip_dbg.so (make -C miphei-vit_amd/csrc dbg), whose dispatch knobs read
    # the environment (MVIT_GEMM_WS, MVIT_GEMM_WS_PF, ...): lets a kernel variant be validated before it becomes the product default
    from miphei_vit_amd import _lib as _mvit_lib
    _mvit_lib.LIB_PATH = _mvit_lib.LIB_PATH.replace("libmiphei_hip.so", "libmiphei_hip_dbg.so")


if os.environ.get("MIPHEI_LIB"):
    # measurement runs only: the same tests against a compile-time variant of the library (make BUILD=... LIB=... EXTRA=-D...)
    from miphei_vit_amd import _lib as _mvit_lib2
    _mvit_lib2.LIB_PATH = os.path.abspath(os.environ["MIPHEI_LIB"])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
