#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2e4
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r2e4/pytest.log
ETH_KZG_AMD_TRACE=1 python tools/bench_abi.py 2048 > gpurun_out/r2e4/abi_trace_full.log 2>&1
grep "host-batch" gpurun_out/r2e4/abi_trace_full.log | grep -v " 1 blobs" | tail -12 > gpurun_out/r2e4/abi_trace.log
python tools/bench_abi.py 2048 2>&1 | grep -v amdgpu.ids > gpurun_out/r2e4/abi.log
python tools/bench_abi.py 512 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2e4/abi.log
python tools/bench_abi.py 5000 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/r2e4/abi.log
cat gpurun_out/r2e4/pytest.log gpurun_out/r2e4/abi_trace.log gpurun_out/r2e4/abi.log
