#!/bin/bash
# ON THE GPU BOX: the in-tree library against the build with the constant factor of the m * p multiply-adds first (-DFP30_SWAP_VP),
# alternating processes on one box; prints blobs/s and the MSM / linear-map stage times of each run.
REPO=$(cd "$(dirname "$0")/.." && pwd)
for round in 1 2 3; do
  for lib in "" "$REPO/tools/ab/libc_eth_kzg_swapvp.so"; do
    if [ -n "$lib" ]; then export ETH_KZG_AMD_LIB=$lib; else unset ETH_KZG_AMD_LIB; fi
    python3 "$REPO/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe 2>/dev/null | python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); s=d['stage_ms_per_step']
print('${lib:+swap_vp }${lib:-in-tree }', round(d['value']), 'blobs/s  msm', s['msm_fixed'], ' linmap', s['g1_linmap'], ' fr', round(s['blob_to_coeffs']+s['coeffs_to_cells']+s['fk20_scalars'],3))"
  done
done
