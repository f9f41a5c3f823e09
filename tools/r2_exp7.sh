#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2e7
ETH_KZG_AMD_TRACE=1 python -c "
import importlib,sys
sys.path.insert(0,'tests')
kzg=importlib.import_module('rust-eth-kzg_amd')
c=kzg.DASContext(True)
print('glv', c.glv_table(), c.window_bits(), c.table_bytes()/1e9)
import synth
from oracle_lib import Oracle
o=Oracle(True,8)
for n in (1,3,9,40,300):
    blobs=[synth.seeded_blob(i) for i in range(n)]
    st,cells,proofs=c.compute_cells_and_kzg_proofs_batch(blobs)
    ec,ep=o.compute_cells_and_kzg_proofs(blobs[n//2])
    print('n',n,'vs oracle:', cells[n//2]==ec, proofs[n//2]==ep)
" > gpurun_out/r2e7/first.log 2>&1
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2e7/pytest.log
for B in 1 64 256 512 1024 2048 4096; do
    python bench.py --blobs-per-gpu $B --steps 5 --warmup 2 --no-cpu-baseline --no-latency-probe --no-configs 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print($B, round(d['value']), round(d['ms_per_step'],2), d['stage_ms_per_step'])"
done > gpurun_out/r2e7/sweep.log 2>&1
ETH_KZG_AMD_WINDOW=14 python bench.py --blobs-per-gpu 2048 --steps 5 --warmup 2 --no-cpu-baseline --no-latency-probe --no-configs 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain14', round(d['value']), round(d['ms_per_step'],2), d['stage_ms_per_step'])" >> gpurun_out/r2e7/sweep.log 2>&1
cat gpurun_out/r2e7/first.log gpurun_out/r2e7/pytest.log gpurun_out/r2e7/sweep.log
