#!/bin/bash
# usage: scripts/gpu_variants.sh name1 name2 ...   ("base" = librt_hip.so); prints one line per variant
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then so=librt_hip.so; else so=librt_hip_$v.so; fi
  RT_HIP_SO=$so timeout -k 10 200 python3 -u bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/var_$v.log 2>&1 || { echo "$v FAILED"; tail -3 gpurun_out/var_$v.log; continue; }
  python3 - "$v" <<'PY'
import json,sys
v=sys.argv[1]
d=json.loads([l for l in open(f'gpurun_out/var_{v}.log') if l.startswith('{')][0])
s=d['roofline']['stage_ms_per_step']
print(f"{v:10s} Mrays/s={d['value']:8.1f} ms/frame={d['ms_per_step']:.3f} closest={s['closest']:.3f} shadow={s['shadow']:.3f} shade={s['shade']:.3f} raygen={s['raygen']:.3f}")
PY
done
