#!/bin/bash
# usage: scripts/gpu_occ.sh 1 2 3 4 ; bench with the persistent trace grids capped at N workgroups per CU
cd $GRAFT_REPO_ROOT
for n in "$@"; do
  RT_TRACE_BLOCKS_PER_CU=$n timeout -k 10 200 python3 -u bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/occ_$n.log 2>&1 || { echo "$n FAILED"; tail -3 gpurun_out/occ_$n.log; exit 1; }
  python3 - "$n" <<'PY'
import json,sys
v=sys.argv[1]
d=json.loads([l for l in open(f'gpurun_out/occ_{v}.log') if l.startswith('{')][0])
s=d['roofline']['stage_ms_per_step']
print(f"blocks/CU={v:3s} Mrays/s={d['value']:8.1f} ms/frame={d['ms_per_step']:.3f} closest={s['closest']:.3f} shadow={s['shadow']:.3f}")
PY
done
