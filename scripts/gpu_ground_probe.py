#!/usr/bin/env python3
"""What does the 2-triangle ground MESH of the bench scene cost each ray?  The same frame with the ground as a plane
(tested before the traversal, raytracer.cpp:423-435) and with no ground at all, stage times of the per-bounce pipeline.
usage: RT_HIP_SO=... gpu_ground_probe.py"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from raytracer_amd import DeviceScene, Stats, RT_OPT_FLAGS, RT_FLAG_TIMING, scenes

def variants():
    s = scenes.mesh_scene(709, 1920, 1080); yield "ground = 2-triangle mesh", s
    s = scenes.mesh_scene(709, 1920, 1080, name="mesh_plane")
    s.meshes = s.meshes[1:]
    s.planes = [dict(point=1, normal=(0, 1, 0), material=2)]
    yield "ground = plane", s
    s = scenes.mesh_scene(709, 1920, 1080, name="mesh_noground")
    s.meshes = s.meshes[1:]
    yield "no ground", s

for name, spec in variants():
    hs = spec.build()
    cam = hs.camera(0)
    ds = DeviceScene(hs, 0)
    ds.set_option(RT_OPT_FLAGS, RT_FLAG_TIMING)
    W, H = cam.image_width, cam.image_height
    fb = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    best = None
    for it in range(8):
        st = Stats()
        ds.render_device(cam, 0, 0, W, H, fb.data_ptr(), W * 3, stats=st)
        if best is None or st.ms_total < best.ms_total: best = st
    rays = best.rays_primary + best.rays_secondary + best.rays_shadow
    print(f"{os.environ.get('RT_HIP_SO', 'librt_hip.so'):22s} {name:26s} {best.ms_total:7.3f} ms {rays/best.ms_total/1e3:7.1f} Mrays/s rays {best.rays_primary}+{best.rays_shadow} | closest {best.ms_closest:.3f} shadow {best.ms_shadow:.3f}", flush=True)
    ds.close()
