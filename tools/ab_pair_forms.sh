#!/bin/bash
# alternated: 64- and 32-blob steps with the pair forms of the signed field (default) and of the 14-digit field (ETH_KZG_AMD_ARENA_SIGNED=0)
for r in 1 2 3; do for v in 1 0; do for B in 64 32; do
  ms=$(ETH_KZG_AMD_ARENA_SIGNED=$v python bench.py --blobs-per-gpu $B --steps 40 --warmup 5 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe --no-device-list-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step_without_stage_events'],3), round(d['ms_per_step'],3), d['stage_ms_per_step']['g1_linmap'], d['stage_ms_per_step']['msm_fixed'])")
  echo "round $r signed=$v blobs=$B: ms/step(no events) ms/step linmap msm = $ms"
done; done; done
