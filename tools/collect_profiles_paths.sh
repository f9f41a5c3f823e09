#!/bin/bash
# Run ON THE GPU BOX (gpurun -- bash tools/collect_profiles_paths.sh TAG): rocprofv3 kernel-trace summaries of the
# verification, recovery and EIP-4844 paths behind profiles/<TAG>_{verify,recover,4844}_kernel_stats.csv.
set -u
TAG=${1:-r2}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for P in verify verify_many recover 4844 single; do
  rocprofv3 --kernel-trace --stats -d "$OUT/trace_$P" -o t --output-format csv -- python3 "$REPO/tools/profile_paths.py" $P > "$OUT/${TAG}_${P}_host_timings.json" 2> "$OUT/trace_$P.err"
  cp "$(find "$OUT/trace_$P" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_${P}_kernel_stats.csv"
  rm -rf "$OUT/trace_$P"
done
ls -la "$OUT"; cat "$OUT"/*_host_timings.json
