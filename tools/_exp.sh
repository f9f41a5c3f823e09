#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "recover" 2>&1 | tail -2
