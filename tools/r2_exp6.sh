#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2e6
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r2e6/pytest.log
( time python bench.py ) > gpurun_out/r2e6/bench.json 2> gpurun_out/r2e6/bench.err
cat gpurun_out/r2e6/pytest.log; tail -5 gpurun_out/r2e6/bench.err; python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r2e6/bench.json') if l.startswith('{')][-1])
print(json.dumps({k:v for k,v in d.items() if k not in ('cpu_baseline',)}, indent=1)[:6000])
print(json.dumps(d.get('cpu_baseline'), indent=1)[:2500])
"
