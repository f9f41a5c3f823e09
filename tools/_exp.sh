#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
for B in 64 2048; do
    python bench.py --blobs-per-gpu $B --steps 5 --warmup 2 --no-cpu-baseline --no-latency-probe --no-configs 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print($B, round(d['value']), round(d['ms_per_step'],2), d['stage_ms_per_step'])"
done
