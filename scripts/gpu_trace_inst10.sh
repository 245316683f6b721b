#!/bin/bash
# per-launch durations of the inst10 1080p frame for library variants: scripts/gpu_trace_inst10.sh "<suffixes>"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in $1; do
  [ "$v" = "''" ] && v=""
  rm -rf /tmp/tr$v
  RT_HIP_SO=librt_hip$v.so RT_QUICK_ITERS=4 timeout -s KILL 150 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$v -- python3 $R/scripts/gpu_quick.py "inst10 1080p" > /tmp/tr$v.log 2>&1 || { echo "trace failed $v"; tail -3 /tmp/tr$v.log; exit 1; }
  echo "== librt_hip$v.so"
  python3 $R/scripts/trace_launches.py /tmp/tr$v | grep k_trace
done
