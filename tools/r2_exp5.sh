#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2e5
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r2e5/pytest.log
bash tools/collect_profiles_paths.sh r2b > gpurun_out/r2e5/paths.log 2>&1
cat gpurun_out/r2e5/pytest.log; tail -8 gpurun_out/r2e5/paths.log
