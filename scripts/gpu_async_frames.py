#!/usr/bin/env python3
"""Wall-clock per frame of the bench workload (mesh1m 1080p, depth 0) for three ways of calling the library: with rt_stats per
frame (synchronous, kernels bracketed by events: what bench.py does), without stats but with RT_FLAG_TIMING, and fully
asynchronous (no stats, no events: frames queue back to back on the stream)."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from raytracer_amd import DeviceScene, Stats, RT_OPT_FLAGS, RT_FLAG_TIMING, scenes
hs = scenes.mesh_scene(709, 1920, 1080).build()
cam = hs.camera(0)
ds = DeviceScene(hs, 0)
W, H = 1920, 1080
fb = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
K = 200
for label, flags, use_stats in (("stats per frame + timing events", RT_FLAG_TIMING, True), ("stats per frame, no events", 0, True), ("asynchronous, no stats, no events", 0, False)):
    ds.set_option(RT_OPT_FLAGS, flags)
    for _ in range(5):
        ds.render_device(cam, 0, 0, W, H, fb.data_ptr(), W * 3, stats=Stats() if use_stats else None, stream=stream)
    ds.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        ds.render_device(cam, 0, 0, W, H, fb.data_ptr(), W * 3, stats=Stats() if use_stats else None, stream=stream)
    ds.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    print(f"{label:40s} {dt:7.4f} ms/frame", flush=True)
ds.close()
