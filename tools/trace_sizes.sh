#!/bin/bash
# Run ON THE GPU BOX (gpurun -- bash tools/trace_sizes.sh 2049 2080 ...): the rocprofv3 kernel-stats summary of the device-resident
# step at each of the given batch sizes -> gpurun_out/trace_sizes/kernel_stats_b<B>.csv (which kernel a size's extra time sits in).
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/trace_sizes
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for B in "$@"; do
  rocprofv3 --kernel-trace --stats -d "$OUT/t$B" -o t --output-format csv -- python3 "$REPO/bench.py" --blobs-per-gpu $B --steps 6 --warmup 2 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe --no-device-list-leg > "$OUT/bench_b$B.json" 2> "$OUT/t$B.err"
  cp "$(find "$OUT/t$B" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats_b$B.csv"
  rm -rf "$OUT/t$B"
  echo "== $B"; head -8 "$OUT/kernel_stats_b$B.csv" | cut -c1-150
done
