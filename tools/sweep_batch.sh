#!/bin/bash
# batch-size curve with per-stage HIP-event times: bash tools/sweep_batch.sh [sizes...]
SIZES=${@:-1 16 64 128 256 512 2048}
for B in $SIZES; do python bench.py --blobs-per-gpu $B --steps 5 --warmup 2 --no-cpu-baseline --no-latency-probe --no-configs 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print($B, round(d['value']), round(d['ms_per_step'],2), d['stage_ms_per_step'])"; done
