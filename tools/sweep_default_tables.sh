#!/bin/bash
# Run ON THE GPU BOX: device-resident steps at small and middle batch sizes on the library's DEFAULT tables (108 GB: nine windows) next
# to the widest ones (bench.py's setting)
for T in 108 max; do for B in ${@:-1 8 16 32 48 56 64 128 256 512}; do
  ms=$(ETH_KZG_AMD_TABLE_GB=$T python bench.py --blobs-per-gpu $B --steps 20 --warmup 3 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe --no-device-list-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print(round(d['ms_per_step_without_stage_events'],3), s['msm_fixed'], s['g1_linmap'], d['config'].get('window_bits'))")
  echo "tables=$T blobs=$B: step / msm / linmap / window bits = $ms"
done; done
