#!/bin/bash
for v in 0 1; do for B in 12 16 20 24 28 32 40 48 64; do
  ms=$(ETH_KZG_AMD_MSM_SPLIT=$v python bench.py --blobs-per-gpu $B --steps 40 --warmup 5 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe --no-device-list-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step_without_stage_events'],3), round(d['ms_per_step'],3), d['stage_ms_per_step']['msm_fixed'])")
  echo "split=$v blobs=$B: no-events / events / msm = $ms"
done; done
