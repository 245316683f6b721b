#!/usr/bin/env python3
"""Per-launch durations of the last frame in a rocprofv3 --kernel-trace CSV: trace_launches.py <dir>"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last frame = from the last ray-generation launch on (a frame of one chunk)
last = max(i for i, r in enumerate(rows) if "k_raygen" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
for r in rows[last:]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  {n}")
