#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
