#!/bin/bash
# Run ON THE GPU BOX: where the circulant form (k_g1circ.hip) hands over to the compiled linear map, alternated on one box.
# Before: tools/build_variant.sh c2 "-DKZG_CIRC_MAX=2" engine.hip   (and c3 likewise)
REPO=$(cd "$(dirname "$0")/.." && pwd)
for r in 1 2 3; do for V in base c2 c3; do
  lib=$REPO/rust-eth-kzg_amd/libc_eth_kzg.so; [ $V != base ] && lib=$REPO/rust-eth-kzg_amd/ab/libc_eth_kzg_$V.so
  for B in 3 4; do
    ms=$(ETH_KZG_AMD_LIB=$lib python $REPO/bench.py --blobs-per-gpu $B --steps 40 --warmup 5 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe --no-device-list-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step_without_stage_events'],3))")
    echo "round $r $V blobs=$B: $ms ms"
  done
done; done
