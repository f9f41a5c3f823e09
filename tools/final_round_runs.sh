#!/bin/bash
# Run ON THE GPU BOX at the end of a round: the GPU suite, the profiles behind bench.py's roofline object, the default bench line and the
# batch-size sweeps, all from one box -> gpurun_out/final/
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/final
mkdir -p "$OUT"
cd "$REPO"
python -m pytest tests -m gpu -x -q > "$OUT/gpu_suite.log" 2>&1; echo "pytest rc=$?" >> "$OUT/gpu_suite.log"
bash tools/collect_profiles.sh r6 > "$OUT/collect.log" 2>&1
cd "$REPO"
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
bash tools/sweep_tiny_and_other.sh > "$OUT/sweep_tiny_and_other.log" 2>&1
bash tools/sweep_small_batches.sh > "$OUT/sweep_small_batches.log" 2>&1
bash tools/sweep_mid_batches.sh > "$OUT/sweep_mid_batches.log" 2>&1
bash tools/sweep_partial_group.sh 33 36 40 48 65 72 2049 > "$OUT/sweep_partial_group.log" 2>&1
tail -3 "$OUT/gpu_suite.log"
