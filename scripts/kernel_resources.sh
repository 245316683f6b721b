#!/bin/bash
# Register / LDS / scratch use of every kernel of librt_hip (compile only; runs on the CPU box).
#   scripts/kernel_resources.sh [extra -D flags]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-gpu-rdc -Iinclude \
  -Iraytracer_amd/csrc "$@" -Rpass-analysis=kernel-resource-usage -o /tmp/librt_hip_res.so raytracer_amd/csrc/rt_hip.hip 2>&1 |
  grep "remark:" | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/Function Name:/{name=$NF} /TotalSGPRs:/{s=$NF} / VGPRs:/{v=$NF} /ScratchSize/{sc=$NF} /Occupancy/{o=$NF} /SGPRs Spill/{ss=$NF} /VGPRs Spill/{vs=$NF} /LDS Size/{printf "%-60.60s vgpr %3s sgpr %3s scratch %4s occ %s spill s%s v%s lds %s\n", name, v, s, sc, o, ss, vs, $NF}'
