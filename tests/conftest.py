import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

def pytest_addoption(parser):
    # measurement runs only: the same tests against a variant build of the library (make -C miphei-vit_amd/csrc dbg, or BUILD=... LIB=variants/...
    # EXTRA=-D...).  An explicit option, never the environment: a stray variable cannot put the suite on another binary.
    parser.addoption("--variant-lib", default=None, help="path of a variant libmiphei_hip build to run the tests against")


def pytest_configure(config):
    variant = config.getoption("--variant-lib")
    if variant:
        from miphei_vit_amd import _lib
        if not os.path.isfile(variant):
            raise pytest.UsageError(f"--variant-lib {variant}: no such file")
        _lib.LIB_PATH = os.path.abspath(variant)
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
