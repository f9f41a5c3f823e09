#!/bin/bash
# Run ON THE GPU BOX: the shader clock and the package power the SMI reports while bench.py's 2048-blob steps run
# (the point kernels are power-limited: the clock they get, not the instruction count alone, sets their time).
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$REPO/gpurun_out/clocks.log}
python3 "$REPO/bench.py" --steps 400 --warmup 3 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe > "$OUT.bench.json" 2> /dev/null &
PID=$!
sleep 14
for i in $(seq 1 40); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' ' >> "$OUT"
  echo >> "$OUT"
  sleep 0.25
  kill -0 $PID 2>/dev/null || break
done
wait $PID
tail -25 "$OUT"
python3 -c "
import json,sys
d=json.load(open('$OUT.bench.json'));print(d['value'],d['ms_per_step'],d['stage_ms_per_step'])"
