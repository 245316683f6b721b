#!/usr/bin/env python3
"""Two uploads of the 1 M-triangle bench scene with the mesh BVHs rebuilt on the device (binned SAH, rt_dsah.h), for
rocprofv3 --kernel-trace --stats: which kernels the build spends its time in.  usage: gpu_dsah_trace.py [device|lbvh]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd import scenes
mode = sys.argv[1] if len(sys.argv) > 1 else "device"
hs = scenes.mesh_scene(709, 1920, 1080).build()
for _ in range(3):
    t = time.perf_counter()
    ds = rt.DeviceScene(hs, 0, rebuild_mesh_bvh=mode)
    print(f"upload ({mode}) {1e3 * (time.perf_counter() - t):.1f} ms", flush=True)
    ds.close()
