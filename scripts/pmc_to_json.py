#!/usr/bin/env python3
"""Output of scripts/profile.sh (one rocprofv3 --kernel-trace --stats run and one --pmc run per pass, all of
bench.py --no-extras = the mesh1m 1920x1080 frame only, all on one build)
-> profiles/pmc_traffic.json: per trace kernel the mean counter values per launch, the average launch time of the
kernel trace, the derived figures bench.py quotes in its `roofline` object, and the hash of the kernel sources they
were taken on (bench.py quotes them only while raytracer_amd/csrc still hashes to it);
-> <kept>_kernel_stats.csv, <kept>_pmc_summary.txt, <kept>_bench_line_under_rocprof.json: the copies kept under profiles/.

usage: scripts/pmc_to_json.py <gpurun_out/name (holds trace/, pass*/)> <profiles/rNN/label>

Derived figures and their units (MI355X_MICROARCH.md, HBM / rocprofv3 sections):
  hbm_bytes_per_launch  FETCH_SIZE [KiB] x 1024 x 2: gfx950 tallies 128-B fabric requests at 64 B.  Infinity-Cache hits
                        are included, so this bounds HBM bytes from above.
  clk_cycles            GRBM_GUI_ACTIVE / 8: rocprofv3 reports the sum over the 8 XCDs.
  valu_issue_util       SQ_INSTS_VALU x 4 cycles (a wave64 instruction occupies its SIMD's 16 lanes for 4 cycles)
                        / (1024 SIMDs x clk_cycles).
  waves_per_simd        SQ_WAVE_CYCLES x 4 (counted in quad-cycles) / clk_cycles / 1024 SIMDs: mean resident waves.
  lane_occupancy        SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64): active lanes per issued VALU instruction.
  l2_hit_rate           TCC_HIT / (TCC_HIT + TCC_MISS).
"""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_sha16

src, kept_as = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if k.startswith("void "):
            k = k[5:]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))

# the kernel trace of the same script run: average launch time per kernel
trace_ms = {}
stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    for row in csv.DictReader(open(stats[0])):
        k = row["Name"].split("(")[0]
        if k.startswith("void "):
            k = k[5:]
        trace_ms[k.replace("<false>", "")] = float(row["AverageNs"]) / 1e6
box_hash = None
hp = os.path.join(src, "csrc_sha16.txt")
if os.path.exists(hp):
    box_hash = open(hp).read().strip()
    if box_hash != csrc_sha16():
        sys.exit(f"the profile was taken on sources {box_hash}, the tree is {csrc_sha16()}: not written")

out = {}
for k, counters in acc.items():
    name = k.replace("<false>", "")          # the parity build of a kernel goes by its plain name
    if not name.startswith("k_trace"):
        continue
    m = {c: sum(v) / len(v) for c, v in counters.items()}
    d = {c: round(v, 1) for c, v in sorted(m.items())}
    d["launches_averaged"] = min(len(v) for v in counters.values())
    if "FETCH_SIZE" in m:
        d["hbm_bytes_per_launch"] = m["FETCH_SIZE"] * 1024 * 2
    if "GRBM_GUI_ACTIVE" in m:
        clk = m["GRBM_GUI_ACTIVE"] / 8
        d["clk_cycles"] = clk
        if "SQ_INSTS_VALU" in m:
            d["valu_insts_per_launch"] = m["SQ_INSTS_VALU"]
            d["valu_issue_util"] = m["SQ_INSTS_VALU"] * 4 / (1024 * clk)
        if "SQ_WAVE_CYCLES" in m:
            d["waves_per_simd"] = m["SQ_WAVE_CYCLES"] * 4 / clk / 1024
    if m.get("SQ_ACTIVE_INST_VALU") and "SQ_THREAD_CYCLES_VALU" in m:
        d["lane_occupancy"] = m["SQ_THREAD_CYCLES_VALU"] / (m["SQ_ACTIVE_INST_VALU"] * 64)
    if "TCC_HIT_sum" in m and m["TCC_HIT_sum"] + m.get("TCC_MISS_sum", 0) > 0:
        d["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    if name in trace_ms:
        d["trace_avg_ms"] = trace_ms[name]
    out[name] = d

doc = {"csrc_sha16": csrc_sha16(), "mesh1m_1920x1080": out, "source": f"{kept_as} (rocprofv3 --pmc, one run per pass, scripts/pmc.sh; scripts/pmc_to_json.py)",
       "note": __doc__.split("Derived figures")[1].strip()}
with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
    json.dump(doc, f, indent=1)
import shutil
os.makedirs(os.path.dirname(os.path.join(ROOT, kept_as)), exist_ok=True)
if stats:
    shutil.copy(stats[0], os.path.join(ROOT, kept_as + "_kernel_stats.csv"))
for a, b in (("pmc_summary.txt", "_pmc_summary.txt"), ("trace_bench_line.json", "_bench_line_under_rocprof.json")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(ROOT, kept_as + b))
print(json.dumps({k: {c: v for c, v in d.items() if c[0].islower()} for k, d in out.items()}, indent=1))
