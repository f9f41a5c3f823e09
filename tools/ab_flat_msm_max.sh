#!/bin/bash
# Run ON THE GPU BOX: up to how many blobs the one-block-per-MSM kernel beats two lanes per window.
# Before: tools/build_variant.sh f8 "-DKZG_FLAT_MSM_MAX_SLICES=8" engine_prover.hip engine.hip   (f16 likewise; the product has 12)
REPO=$(cd "$(dirname "$0")/.." && pwd)
for r in 1 2; do for V in ${VARIANTS:-base f8}; do
  lib=$REPO/rust-eth-kzg_amd/libc_eth_kzg.so; [ $V != base ] && lib=$REPO/rust-eth-kzg_amd/ab/libc_eth_kzg_$V.so
  for B in 9 10 12 14 16; do
    ms=$(ETH_KZG_AMD_LIB=$lib python $REPO/bench.py --blobs-per-gpu $B --steps 40 --warmup 5 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe --no-device-list-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step_without_stage_events'],3), d['stage_ms_per_step']['msm_fixed'])")
    echo "round $r $V blobs=$B: step / msm = $ms"
  done
done; done
