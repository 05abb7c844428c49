# Measurement only: socket power / sclk samples (rocm-smi) while bench.py runs ~120 training steps; prints the busy samples
cd "$(dirname "$0")/.."
python3 bench.py --steps 120 --warmup 5 --no-cpu-baseline > /tmp/step_power_bench.json 2>/dev/null &
pid=$!
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 0.25
done | awk '{p=$NF; if (p+0 > 500) print}' | tail -14
cut -c1-140 /tmp/step_power_bench.json
