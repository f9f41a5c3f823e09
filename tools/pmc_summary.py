"""Fold rocprofv3 --pmc counter_collection CSVs (one or more passes) into profiles/<name>.json:
per kernel, the maximum over launches of every counter (a full-batch launch is the maximum) and the launch count.
usage: python tools/pmc_summary.py OUT.json "note" DIR [DIR ...]"""
import csv, glob, json, os, re, sys

out, note, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
kern = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = {}
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\((?!anonymous namespace\)).*", "", r["Kernel_Name"]).strip()  # drop the argument list, keep "(anonymous namespace)"
            key = (name, r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] = per_dispatch.get(key, 0.0) + float(r["Counter_Value"])  # sum over XCDs / instances
        for (name, _, ctr), v in per_dispatch.items():
            k = kern.setdefault(name, {})
            k[ctr + "_per_launch_max"] = max(k.get(ctr + "_per_launch_max", 0.0), v)
            k[ctr + "_launches"] = k.get(ctr + "_launches", 0) + 1
# the key bench.py uses to decide whether these counters belong to the build it is running (bench.py: kernel_sources_hash)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
try:
    from bench import kernel_sources_hash
    src_hash = kernel_sources_hash()
except Exception as e:  # pragma: no cover
    src_hash = None
json.dump({"note": note, "msm_kernel_sources_sha256": src_hash, "kernels": kern}, open(out, "w"), indent=1)
print(out, len(kern), "kernels")
