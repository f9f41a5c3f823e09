#!/bin/bash
# ON THE GPU BOX: gpurun -- bash tools/run_calib_gather96.sh   (plain run for the times, then the FETCH_SIZE pass)
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/calib96
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
"$REPO/tools/calib_gather96" > "$OUT/plain_run.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc" -o p --output-format csv -- "$REPO/tools/calib_gather96" > "$OUT/pmc_run.log" 2>&1
cat "$OUT/plain_run.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for row in csv.DictReader(open(f[0])):
    if row["Counter_Name"] == "FETCH_SIZE":
        acc[(row["Kernel_Name"][:60], row["Dispatch_Id"])].append(float(row["Counter_Value"]))
per = collections.defaultdict(list)
for (k, d), v in acc.items():
    per[k].append(sum(v))
for k, v in per.items():
    print(k, ["%.3f GB (x2: %.3f)" % (x * 1024 / 1e9, 2 * x * 1024 / 1e9) for x in v])
PY
