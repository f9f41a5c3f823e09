#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "verify" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "verify" 2>&1 | tail -5
ETH_KZG_AMD_TRACE=1 python tools/profile_paths.py verify 2>&1 | grep -E "verify\]|verify_ms" | tail -12
ETH_KZG_AMD_PIP_SHIFT_MIN=1000000 ETH_KZG_AMD_TRACE=1 python tools/profile_paths.py verify 2>&1 | grep -E "verify\]|verify_ms" | tail -7
