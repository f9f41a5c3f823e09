#!/bin/bash
# Run ON THE GPU BOX: the one-block-per-MSM kernel with 256 / 128 / 64 threads per block (experiment switch KZG_FLAT_NT), alternated
for r in 1 2; do for NT in 256 128 64  # (the switch existed in the experiment build only: commit 1e5642c + this script); do for B in 1 2 3 4 5 8; do
  ms=$(KZG_FLAT_NT=$NT python bench.py --blobs-per-gpu $B --steps 40 --warmup 5 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe --no-device-list-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print(round(d['ms_per_step_without_stage_events'],3), s['msm_fixed'], s.get('g1_ifft'), s['g1_linmap'])")
  echo "round $r threads=$NT blobs=$B: step / msm / circ / linmap = $ms"
done; done; done
