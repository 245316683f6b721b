#!/usr/bin/env python3
"""Per-kernel totals of ONE device SAH build (the last upload in a rocprofv3 --kernel-trace of scripts/gpu_dsah_trace.py) and the per-level
durations of its two heaviest passes.  usage: dsah_trace_summary.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
starts = [i for i, r in enumerate(rows) if name(r).endswith("k_prims")]
# every upload runs k_prims once per mesh BVH (ground, sphere): the last upload starts at the second-to-last k_prims
first = starts[-2] if len(starts) >= 2 else 0
last = rows[first:]
tot = collections.OrderedDict()
for r in last:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = name(r)
    if "rocprim" in k: k = "rocprim scan kernels"
    t = tot.setdefault(k, [0, 0.0]); t[0] += 1; t[1] += d
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3
print(f"last upload: {len(last)} dispatches over {span/1e3:.2f} ms, busy {sum(v[1] for v in tot.values())/1e3:.2f} ms")
for k, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:40s} x{c:4d} {d/1e3:8.3f} ms")
for kn in ("rt_dsah::k_cbounds", "rt_dsah::k_bin", "rt_dsah::k_scatter"):
    ds = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last if name(r) == kn]
    print(f"  {kn} per level (us): " + " ".join(f"{d:.0f}" for d in ds))
