cd "$(dirname "$0")/../.."
python -m pytest tests/test_gemm_gpu.py -q -x 2>&1 | tail -2
echo "== PIPE8=1"; python tools/bench_vs_blas.py 2>&1 | grep "ours"
echo "== PIPE8=0"; python - <<'PY' 2>&1 | grep ours | sed 's/| hipBLASLt.*//'
import sys, os, runpy
sys.path.insert(0, os.getcwd())
import miphei_vit_amd._lib as L
L.LIB_PATH = os.path.abspath("tools/abl/libmiphei_p80.so")
sys.argv = ["tools/bench_vs_blas.py"]
runpy.run_path("tools/bench_vs_blas.py", run_name="__main__")
PY
