# Measurement only: tools/bench_vs_blas.py through the product library and through every alternative build dropped into tools/abl/*.so
# (e.g. gemm units rebuilt with -DMVIT_ABLATE=n / -DMVIT_GEMM_HO=n), same box, same process order.
cd "$(dirname "$0")/../.."
for l in miphei-vit_amd/libmiphei_hip.so tools/abl/*.so; do echo "== $l"; python - $l <<'PY' 2>&1 | grep ours | sed 's/| hipBLASLt.*//'
import sys, os, runpy
sys.path.insert(0, os.getcwd())
import miphei_vit_amd._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["tools/bench_vs_blas.py"]
runpy.run_path("tools/bench_vs_blas.py", run_name="__main__")
PY
done
