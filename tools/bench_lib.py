"""bench.py on a chosen build of the library (MIPHEI_LIB=path/to/libmiphei_x.so): same-box A/B of compile-time variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miphei_vit_amd import _lib
if os.environ.get("MIPHEI_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIPHEI_LIB"])
import bench
sys.exit(bench.main())
