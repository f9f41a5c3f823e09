#!/bin/bash
# A/B on ONE box: bash tools/ab_compare.sh libA.so libB.so [batch sizes]; alternates the two builds (boxes differ by ~10 %)
A=$1; B=$2; shift 2
for rep in 1 2; do for L in "$A" "$B"; do echo "== $L"; ETH_KZG_AMD_LIB=$L bash tools/sweep_batch.sh ${@:-2048}; done; done
