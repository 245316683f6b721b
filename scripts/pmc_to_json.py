#!/usr/bin/env python3
"""rocprofv3 --pmc passes (scripts/pmc.sh: one run per pass, bench.py --no-extras = the mesh1m 1920x1080 frame only)
-> profiles/pmc_traffic.json: per trace kernel the mean counter values per launch, the derived figures bench.py
quotes in its `roofline` object, and the hash of the kernel sources they were taken on (bench.py quotes them only
while raytracer_amd/csrc still hashes to it).

usage: scripts/pmc_to_json.py <dir with pass*/> <profiles/rNN/name of the summary kept beside it>

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
    out[name] = d

doc = {"csrc_sha16": csrc_sha16(), "mesh1m_1920x1080": out, "source": f"{kept_as} (rocprofv3 --pmc, one run per pass, scripts/pmc.sh; scripts/pmc_to_json.py)",
       "note": __doc__.split("Derived figures")[1].strip()}
with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
    json.dump(doc, f, indent=1)
print(json.dumps(doc, indent=1))
