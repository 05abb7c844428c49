cd "$(dirname "$0")/../.."
for l in miphei-vit_amd/libmiphei_hip.so tools/abl/*.so miphei-vit_amd/libmiphei_hip.so; do echo "== $l"; python - $l <<'PY' 2>&1 | grep "^N="
import sys, os, runpy
sys.path.insert(0, os.getcwd())
import miphei_vit_amd._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["tools/bench_attn.py"]
runpy.run_path("tools/bench_attn.py", run_name="__main__")
PY
done
