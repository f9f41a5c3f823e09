#!/bin/bash
# Run ON THE GPU BOX: one PMC pass (SQ_* / GRBM) of bench.py's 2048-blob step, summarised per kernel into gpurun_out/<TAG>_pmc_sq.json
set -u
TAG=${1:-r5}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d "$OUT/pmc_SQ_WAVES" -o p --output-format csv -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-latency-probe --no-configs --no-build-probe > /dev/null 2> "$OUT/pmc_SQ.err"
python3 "$REPO/tools/pmc_summary.py" "$OUT/${TAG}_pmc_sq.json" "SQ pass only" "$OUT/pmc_SQ_WAVES"
rm -rf "$OUT"/pmc_*/
python3 - "$OUT/${TAG}_pmc_sq.json" <<'P'
import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d['kernels'].items():
    if 'chunked' in k or 'mulc' in k: print(k, {a:b for a,b in v.items() if 'launches' not in a})
P
