for r in 1 2; do for B in 2049 2112 2080 2048; do
  ms=$(python bench.py --blobs-per-gpu $B --steps 10 --warmup 3 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe --no-device-list-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print(round(d['ms_per_step_without_stage_events'],3), round(d['ms_per_step'],3), s['msm_fixed'], s['g1_linmap'])")
  echo "blobs=$B: no-events / events / msm / linmap = $ms"
done; done
