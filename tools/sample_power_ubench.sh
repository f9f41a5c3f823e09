#!/bin/bash
# Run ON THE GPU BOX: package power and shader clock while tools/ubench_fp30 --sustained runs (1.5 s per kernel)
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$REPO/gpurun_out/ubench_power.log}
"$REPO/tools/ubench_fp30" --sustained > "$OUT.ubench" 2>&1 &
PID=$!
while kill -0 $PID 2>/dev/null; do
  echo "$(date +%s.%N) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Package Power' | sed -E 's/.*\(([0-9]+)Mhz\).*/sclk \1/; s/.*Power \(W\): ([0-9.]+).*/W \1/' | tr '\n' ' ')" >> "$OUT"
  sleep 0.2
done
cat "$OUT.ubench"; cat "$OUT"
