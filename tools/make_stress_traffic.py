"""Per-launch HBM traffic of the traversal kernels in a tools/profile_stress.sh directory: joins the kernel trace
(durations) with the FETCH_SIZE / WRITE_SIZE passes by launch order.  FETCH_SIZE is doubled (gfx950 counts 64 B per
128-B fabric read request); the three passes are separate runs of the same deterministic command."""
import csv, json, sys, os
root = sys.argv[1]
def rows(path, pred):
    return [r for r in csv.DictReader(open(path)) if pred(r)]
out = {"run": json.loads([l for l in open(os.path.join(root, "trace.log")) if l.startswith('{"copies"')][-1]), "kernels": {}}
for kern in ("k_trace<false>", "k_shadow<false>", "k_surface"):
    tr = rows(os.path.join(root, "trace", "k_kernel_trace.csv"), lambda r: kern in r["Kernel_Name"])
    fe = rows(os.path.join(root, "fetch", "pmc_counter_collection.csv"), lambda r: kern in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE")
    wr = rows(os.path.join(root, "write", "pmc_counter_collection.csv"), lambda r: kern in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE")
    n = min(len(tr), len(fe), len(wr))
    launches = []
    for k in range(n):
        ms = (int(tr[k]["End_Timestamp"]) - int(tr[k]["Start_Timestamp"])) / 1e6
        b = 2 * float(fe[k]["Counter_Value"]) * 1024 + float(wr[k]["Counter_Value"]) * 1024
        launches.append({"ms": round(ms, 3), "hbm_gb": round(b / 1e9, 3), "tb_per_s": round(b / (ms * 1e-3) / 1e12, 2)})
    out["kernels"][kern] = launches
json.dump(out, open(os.path.join(root, "stress_traffic.json"), "w"), indent=1)
for k, v in out["kernels"].items():
    print(k, [(l["ms"], l["tb_per_s"]) for l in v])
