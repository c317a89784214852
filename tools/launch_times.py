#!/usr/bin/env python3
"""Per-launch durations of the last frame of a rocprofv3 kernel trace (k_kernel_trace.csv), in launch order."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("void ", "").split("(")[0] for r in rows]
# last frame: from the last launch that begins a frame (the last k_trace of a CAM kind or k_raygen) on
last = max(i for i, n in enumerate(names) if "k_raygen" in n or ("k_trace" in n and "true>" in n.split(",")[-1]))
t0 = int(rows[last]["Start_Timestamp"])
for r, n in list(zip(rows, names))[last:]:
    if n.startswith("k_reset"):
        continue
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:9.3f} ms  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:8.3f} ms  {n}")
