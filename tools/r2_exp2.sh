#!/bin/bash
# round-2 experiment 2: compiled linear map for the G1 transforms
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2e2
ETH_KZG_AMD_TRACE=1 python -c "
import importlib,sys
sys.path.insert(0,'tests')
kzg=importlib.import_module('rust-eth-kzg_amd')
c=kzg.DASContext(True)
import synth
from oracle_lib import Oracle
o=Oracle(True,8)
blobs=[synth.seeded_blob(i) for i in range(40)]
st,cells,proofs=c.compute_cells_and_kzg_proofs_batch(blobs)
ec,ep=o.compute_cells_and_kzg_proofs(blobs[7])
print('linmap 40 blobs vs oracle:', cells[7]==ec, proofs[7]==ep)
" > gpurun_out/r2e2/first.log 2>&1
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2e2/pytest.log
for B in 33 64 2048; do
    python bench.py --blobs-per-gpu $B --steps 5 --warmup 2 --no-cpu-baseline --no-latency-probe 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print($B, round(d['value']), round(d['ms_per_step'],2), d['stage_ms_per_step'])"
done > gpurun_out/r2e2/sweep.log 2>&1
ETH_KZG_AMD_NO_TOOM8=1 python bench.py --blobs-per-gpu 2048 --steps 5 --warmup 2 --no-cpu-baseline --no-latency-probe 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('no-toom8', round(d['value']), round(d['ms_per_step'],2), d['stage_ms_per_step'])" >> gpurun_out/r2e2/sweep.log 2>&1
cat gpurun_out/r2e2/first.log gpurun_out/r2e2/pytest.log gpurun_out/r2e2/sweep.log
