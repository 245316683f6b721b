#!/usr/bin/env python3
"""Key fields of a bench.py line: print_bench.py <file with the JSON line>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus", "steps")})
r = d["roofline"]
print("roofline", {k: r.get(k) for k in ("kernel", "frac", "avg_launch_ms", "traffic", "saturated")})
print("binding", r.get("binding"))
print("other", r.get("other_kernel"))
for k in ("near_first", "rebuilt_bvh", "rebuilt_bvh_on_device", "rebuilt_lbvh_on_device"):
    if k in d:
        print(k, round(d[k]["value"], 1), round(d[k]["closest_ms_per_step"], 4), d[k]["pixels_equal_to_exact_frame"])
for k in ("strong_scaling_base", "strong_scaling_projection", "cpu_baseline", "parity_vs_cpu_reference"):
    if k in d:
        print(k, d[k])
