"""Turns the PMC summary of a profile directory into traffic.json (HBM bytes per launch of the closest-hit kernel).

FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-B fabric read request for wide
coalesced streams, i.e. half the bytes; this kernel's reads are 16-B-per-lane gathers, for which the factor is not
calibrated, so both the raw and the doubled figure are kept and the doubled one (the upper bound) is reported."""
import json, sys, os
root = sys.argv[1]
d = json.load(open(os.path.join(root, "pmc_summary.json")))
bench = json.loads(open(os.path.join(root, "bench.json")).read().strip().splitlines()[-1])
out = {"workload": bench["config"]["workload"], "kernels": {}}
for k, v in d.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        f, w = v["FETCH_SIZE"]["mean"] * 1024, v["WRITE_SIZE"]["mean"] * 1024
        out["kernels"][k] = {"fetch_bytes_raw": f, "fetch_bytes_x2": 2 * f, "write_bytes": w, "launches": v["FETCH_SIZE"]["launches"],
                             "l2_hit_rate": v["TCC_HIT_sum"]["mean"] / max(v["TCC_HIT_sum"]["mean"] + v["TCC_MISS_sum"]["mean"], 1)}
kt = [k for k in out["kernels"] if "k_trace" in k]
if kt:
    t = out["kernels"][kt[0]]
    out["hbm_bytes_per_launch"] = t["fetch_bytes_x2"] + t["write_bytes"]
json.dump(out, open(os.path.join(root, "traffic.json"), "w"), indent=1)
print(json.dumps(out.get("hbm_bytes_per_launch")), kt)
