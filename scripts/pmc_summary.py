"""Summarises the rocprofv3 --pmc passes of scripts/pmc.sh (gpurun_out/pmc/*) per kernel -> JSON.
Usage: python scripts/pmc_summary.py gpurun_out/pmc out.json"""
import collections, csv, glob, json, os, sys
root, out = sys.argv[1], sys.argv[2]
res = {}
for name in ["fetch", "write", "tcc", "sq1", "sq2"]:
    fs = sorted(glob.glob(os.path.join(root, name, "*", "*counter_collection.csv")), key=os.path.getmtime)
    if not fs:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[-1])):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        for c, v in d.items():
            res.setdefault(k, {})[c] = sum(v) / len(v)
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for k in sorted(res):
    if k.startswith("k_"):
        d = res[k]
        print("%-18s FETCH %10.0f KB  WRITE %10.0f KB  wave cycles %.3g  wait_any %.3g" % (k, d.get("FETCH_SIZE", 0), d.get("WRITE_SIZE", 0), d.get("SQ_WAVE_CYCLES", 0), d.get("SQ_WAIT_ANY", 0)))
