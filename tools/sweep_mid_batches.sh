#!/bin/bash
for B in 72 80 96 112 128 160 192 224 256 257 288 320 384; do
  ms=$(python bench.py --blobs-per-gpu $B --steps 20 --warmup 3 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe --no-device-list-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print(round(d['ms_per_step_without_stage_events'],3), round(d['ms_per_step'],3), s['msm_fixed'], s['g1_linmap'], s['coeffs_to_cells'])")
  echo "blobs=$B: no-events / events / msm / linmap / cells = $ms"
done
