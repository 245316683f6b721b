#!/bin/bash
# A/B of library variants on the GPU box: scripts/gpu_ab.sh "<variant suffixes, '' = the product>" [gpu_quick.py args]
# e.g. scripts/gpu_ab.sh "'' _a _b" "mesh1m 1080p" ; with RT_AB_PMC=1 also one PMC pass (instruction counts) per variant
R=$GRAFT_REPO_ROOT
V="$1"; shift
for v in $V; do
  [ "$v" = "''" ] && v=""
  RT_HIP_SO=librt_hip$v.so timeout -k 10 200 python3 $R/scripts/gpu_quick.py "$@" || exit 1
done
if [ -n "$RT_AB_PMC" ]; then
  cd /tmp && export TMPDIR=/tmp
  for v in $V; do
    [ "$v" = "''" ] && v=""
    rm -rf /tmp/pmc$v
    RT_HIP_SO=librt_hip$v.so timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_SMEM --output-format csv -d /tmp/pmc$v -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extras > /tmp/pmc$v.log 2>&1 || { echo "pmc failed for $v"; tail -3 /tmp/pmc$v.log; exit 1; }
    python3 - "$v" <<'PY'
import csv, glob, sys, collections
v = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"/tmp/pmc{v}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    if k.startswith("k_trace"):
        m = {c: sum(x) / len(x) for c, x in acc[k].items()}
        occ = m["SQ_THREAD_CYCLES_VALU"] / (m["SQ_ACTIVE_INST_VALU"] * 64) if m.get("SQ_ACTIVE_INST_VALU") else 0
        print(f"pmc librt_hip{v}.so {k:28s} VALU {m['SQ_INSTS_VALU']/1e6:7.1f}M SALU {m['SQ_INSTS_SALU']/1e6:7.1f}M LDS {m['SQ_INSTS_LDS']/1e6:6.1f}M VMEM_RD {m['SQ_INSTS_VMEM_RD']/1e6:6.1f}M BRANCH {m.get('SQ_INSTS_BRANCH',0)/1e6:6.1f}M SMEM {m.get('SQ_INSTS_SMEM',0)/1e6:6.1f}M lane-occ {occ:.3f}")
PY
  done
fi
