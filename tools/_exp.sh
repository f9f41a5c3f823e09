#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "verify or decompress or subgroup or g1" 2>&1 | tail -3
ETH_KZG_AMD_TRACE=1 python tools/profile_paths.py single 2>&1 | grep -E "verify\]|path" | tail -5
ETH_KZG_AMD_TRACE=1 python tools/profile_paths.py verify 2>&1 | grep -E "verify\]|path" | tail -5
