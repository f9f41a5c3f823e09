#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/bench_final.json') if l.startswith('{')][-1])
print(round(d['value']), round(d['ms_per_step'],2))
print({k:(v.get('blobs_per_s') or v.get('ms') or v.get('commitments_per_s')) for k,v in d['configs'].items() if isinstance(v,dict) and k!='context_creation_s'})
"
