#!/bin/bash
cd "$(dirname "$0")/.."
ETH_KZG_AMD_TRACE=1 python tools/bench_abi.py 2048 2>&1 | grep "host-batch" | grep -v "(1 blobs)\|of 1 blobs\|] 1 blobs" | tail -11
