"""Summarises rocprofv3 --pmc CSVs (one row per dispatch and counter) per kernel: mean counter value per launch."""
import csv, glob, sys, collections, json
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in agg.items():
    out[k] = {c: {"launches": len(v), "mean": sum(v) / len(v), "sum": sum(v)} for c, v in d.items()}
json.dump(out, open(root + "/pmc_summary.json", "w"), indent=1)
for k, d in sorted(out.items()):
    if not k.startswith(("k_", "void k_")): continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} mean/launch {v['mean']:.4g}  launches {v['launches']}")
