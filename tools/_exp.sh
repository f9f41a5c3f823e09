#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "compute or regime or batch" 2>&1 | tail -2
python tools/profile_paths.py single 2>&1 | tail -1
python bench.py --blobs-per-gpu 1 --steps 20 --no-configs --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
python bench.py --blobs-per-gpu 4 --steps 20 --no-configs --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
