#!/bin/bash
# device-resident steps of 1 .. 12 blobs, and 1900 .. 2100 around the bench batch (ms per step without / with the per-stage events)
for B in 1 2 3 4 5 6 8 9 10 12 1900 1984 2000 2047 2048 2049 2112; do
  ms=$(python bench.py --blobs-per-gpu $B --steps 12 --warmup 3 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe --no-device-list-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print(round(d['ms_per_step_without_stage_events'],3), round(d['ms_per_step'],3), s['msm_fixed'], s['g1_linmap'], s['g1_ifft'], round(d['value']))")
  echo "blobs=$B: no-events / events / msm / linmap / circ / blobs-per-s = $ms"
done
