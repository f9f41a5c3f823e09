#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2e3
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2e3/pytest.log
python tools/bench_abi.py 2048 > gpurun_out/r2e3/abi.log 2>&1
python tools/bench_abi.py 512 >> gpurun_out/r2e3/abi.log 2>&1
python tools/bench_abi.py 64 >> gpurun_out/r2e3/abi.log 2>&1
cat gpurun_out/r2e3/pytest.log gpurun_out/r2e3/abi.log
