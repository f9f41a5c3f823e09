#!/bin/bash
# alternated: small-batch steps with two lanes per MSM window (default) and one (ETH_KZG_AMD_MSM_SPLIT=0)
for r in 1 2 3; do for v in 1 0; do for B in 32 16 24; do
  ms=$(ETH_KZG_AMD_MSM_SPLIT=$v python bench.py --blobs-per-gpu $B --steps 40 --warmup 5 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe --no-device-list-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step_without_stage_events'],3), round(d['ms_per_step'],3), d['stage_ms_per_step']['msm_fixed'], d['stage_ms_per_step']['g1_linmap'])")
  echo "round $r split=$v blobs=$B: ms/step(no events) ms/step msm linmap = $ms"
done; done; done
