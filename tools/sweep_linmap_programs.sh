#!/bin/bash
# every compilation of the G1 linear map (ETH_KZG_AMD_SLP_PROGRAM) at every batch size: bash tools/sweep_linmap_programs.sh [sizes...]
SIZES=${@:-32 64 128 192 256 320 384 512}
for P in default 1 2 3 4 5; do
  echo "program $P"
  if [ "$P" = default ]; then unset ETH_KZG_AMD_SLP_PROGRAM; else export ETH_KZG_AMD_SLP_PROGRAM=$P; fi
  bash "$(dirname "$0")/sweep_batch.sh" $SIZES
done
