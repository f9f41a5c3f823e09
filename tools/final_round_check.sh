#!/bin/bash
# Run ON THE GPU BOX after the last commit of a round: the GPU suite, smoke, the default bench line (with the PMC traffic of the committed
# profile quoted) and the path profiles -> gpurun_out/final/
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/final
mkdir -p "$OUT"
cd "$REPO"
python -m pytest tests -m gpu -x -q > "$OUT/gpu_suite.log" 2>&1; echo "pytest rc=$?" >> "$OUT/gpu_suite.log"
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" >> "$OUT/smoke.log"
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
bash tools/collect_profiles_paths.sh r6 > "$OUT/paths.log" 2>&1
cd "$REPO"
tail -3 "$OUT/gpu_suite.log"; tail -2 "$OUT/smoke.log"
