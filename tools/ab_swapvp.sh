#!/bin/bash
# ON THE GPU BOX: the in-tree library against a second build of it, alternating processes on one box; prints blobs/s and the MSM /
# linear-map stage times of each run.  Round 5 used it for the order of the multiply-adds' factors (the second build then had the
# modulus digit first; today that is the default and -DFP30_VP_M_FIRST gives the old order).  Build the second library HERE first:
#   make -C rust-eth-kzg_amd/csrc -j8 HIPCC="hipcc -DFP30_VP_M_FIRST" OBJDIR=/tmp/build_ab OUT=$PWD/tools/ab/libc_eth_kzg_swapvp.so STATIC=/tmp/build_ab/lib.a
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
