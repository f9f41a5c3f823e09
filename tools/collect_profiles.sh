#!/bin/bash
# Run ON THE GPU BOX (gpurun -- bash tools/collect_profiles.sh TAG): the kernel-trace summary and the three PMC passes
# behind profiles/<TAG>_*.  rocprofv3 gets the program directly after `--` (no env / bash -c hops).
set -u
TAG=${1:-r3}
SUFFIX=${2:-glv16}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t --output-format csv -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
for pass in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  d="$OUT/pmc_$(echo $pass | cut -d' ' -f1)"
  rocprofv3 --pmc $pass -d "$d" -o p --output-format csv -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe > /dev/null 2> "$d.err"
done
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats_bench_b2048_${SUFFIX}.csv"
grep '^{' "$OUT/bench_under_rocprof.json" > "$OUT/${TAG}_bench_b2048_${SUFFIX}_under_rocprof.json"
python3 "$REPO/tools/pmc_summary.py" "$OUT/${TAG}_pmc_b2048_${SUFFIX}.json" "rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_*/GRBM in a third pass), bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe (2048 blobs, default FK20 table); counter values summed over XCDs per dispatch, then the maximum over launches; FETCH_SIZE/WRITE_SIZE in KB as reported" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$OUT/pmc_SQ_WAVES"
# single-blob (latency) regime
rocprofv3 --kernel-trace --stats -d "$OUT/trace_b1" -o t --output-format csv -- python3 "$REPO/bench.py" --blobs-per-gpu 1 --steps 20 --warmup 2 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe > "$OUT/bench_b1.json" 2> "$OUT/trace_b1.err"
cp "$(find "$OUT/trace_b1" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats_bench_b1.csv"
# the batches that do not fill the chip (BASELINE configs 4 and 5 per GPU, round 4's subject): 64 and 256 blobs
for B in 64 256; do
  rocprofv3 --kernel-trace --stats -d "$OUT/trace_b$B" -o t --output-format csv -- python3 "$REPO/bench.py" --blobs-per-gpu $B --steps 20 --warmup 2 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe > "$OUT/bench_b$B.json" 2> "$OUT/trace_b$B.err"
  cp "$(find "$OUT/trace_b$B" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats_bench_b$B.csv"
  grep '^{' "$OUT/bench_b$B.json" > "$OUT/${TAG}_bench_b${B}_under_rocprof.json"
  rm -rf "$OUT/trace_b$B"
done
rm -rf "$OUT"/trace "$OUT"/trace_b1 "$OUT"/pmc_*/ 2>/dev/null
ls -la "$OUT"
