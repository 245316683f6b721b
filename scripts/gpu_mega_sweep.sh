#!/bin/bash
timeout -k 10 120 python -u scripts/gpu_mega_probe.py small 2>&1 | grep -E "oracle|mega == bounce" | cut -c1-110 || exit 1
for cfg in "0 8" "0 4" "0 0" "0 11"; do
  set -- $cfg
  echo "== strict $1 any_waves $2"
  RT_MEGA_STRICT=$1 RT_MEGA_ANY_WAVES=$2 timeout -k 10 200 python -u scripts/gpu_mega_probe.py big 2>&1 | grep -E " mega  | bounce  |False" | cut -c1-100 || exit 1
done
