#!/bin/bash
# round-2 experiment 1: chunked MSM kernel vs the windowed one
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2e1
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r2e1/pytest.log
for S in 0 1 2 4; do
  for B in 2048; do
    ETH_KZG_AMD_MSM_CHUNKS=$S python bench.py --blobs-per-gpu $B --steps 5 --warmup 2 --no-cpu-baseline --no-latency-probe 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('S=$S', $B, round(d['value']), round(d['ms_per_step'],2), d['stage_ms_per_step'])"
  done
done > gpurun_out/r2e1/chunks.log 2>&1
for B in 64 128 256 512 1024 1536 3072 4096; do
  for S in 0 4 -1; do
    if [ $S = -1 ]; then unset ETH_KZG_AMD_MSM_CHUNKS; else export ETH_KZG_AMD_MSM_CHUNKS=$S; fi
    python bench.py --blobs-per-gpu $B --steps 5 --warmup 2 --no-cpu-baseline --no-latency-probe 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('S=$S', $B, round(d['value']), round(d['ms_per_step'],2), d['stage_ms_per_step'])"
  done
done >> gpurun_out/r2e1/chunks.log 2>&1
cat gpurun_out/r2e1/pytest.log gpurun_out/r2e1/chunks.log
