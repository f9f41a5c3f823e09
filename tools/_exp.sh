#!/bin/bash
cd "$(dirname "$0")/.."
for B in 4096 8192; do
    python bench.py --blobs-per-gpu $B --steps 4 --warmup 1 --no-cpu-baseline --no-latency-probe --no-configs 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print($B, round(d['value']), round(d['ms_per_step'],2), d['stage_ms_per_step'])"
done
python tools/bench_abi.py 4096 2>&1 | grep -v amdgpu.ids | head -1
python -c "
import __graft_entry__ as g
g.smoke()
"
