#!/bin/bash
# how the stage times move with the number of blobs in the LAST lane group (the batch is padded to a multiple of 64 lanes)
for B in ${@:-2049 2050 2052 2056 2064 2072 2080 2096 2111 2112 129 130 136 160 192}; do
  ms=$(python bench.py --blobs-per-gpu $B --steps 10 --warmup 3 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe --no-device-list-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print(round(d['ms_per_step_without_stage_events'],3), round(d['ms_per_step'],3), s['msm_fixed'], s['g1_linmap'])")
  echo "blobs=$B: no-events / events / msm / linmap = $ms"
done
