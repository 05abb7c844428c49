cd "$(dirname "$0")/../.."
( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.5; done ) > /tmp/pw.txt &
sp=$!
tools/probes/mfma_power 1; RR_ZEROS=1 tools/probes/mfma_power 1
kill $sp
awk '{if ($NF+0 > 400) print}' /tmp/pw.txt | awk 'NR%3==0' | head -12
