#!/usr/bin/env python3
"""BASELINE.json configs[3] at its full size: the instanced depth-6 scene at 3840x2160 with 64 samples per pixel
(531 M primary samples, chunked by the library), exact mode and the two non-parity modes; one frame each."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from raytracer_amd import DeviceScene, Stats, RT_OPT_FLAGS, RT_FLAG_TIMING, RT_FLAG_NEAR_FIRST, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hs = scenes.instanced_scene(709, 3840, 2160, num_samples=spp).build()
cam = hs.camera(0)
fb = torch.zeros((cam.image_height, cam.image_width, 3), dtype=torch.uint8, device="cuda")
ref = None
for mode in ("exact", "near-first", "rebuilt"):
    ds = DeviceScene(hs, 0, rebuild_mesh_bvh=(mode == "rebuilt"))
    ds.set_option(RT_OPT_FLAGS, RT_FLAG_TIMING | (RT_FLAG_NEAR_FIRST if mode == "near-first" else 0))
    st = Stats()
    t = time.perf_counter()
    ds.render_device(cam, 0, 0, cam.image_width, cam.image_height, fb.data_ptr(), cam.image_width * 3, stats=st)
    dt = time.perf_counter() - t
    ds.close()
    img = fb.cpu()
    if ref is None:
        ref = img.clone()
    d = (img.to(torch.int16) - ref.to(torch.int16)).abs().amax(dim=2)
    rays = st.rays_primary + st.rays_secondary + st.rays_shadow
    print(f"{mode:10s} 3840x2160 {spp} spp depth 6: {rays/1e6:8.1f} M rays, {st.ms_total/1e3:6.3f} s/frame (wall {dt:6.3f} s), {rays/st.ms_total/1e3:7.1f} Mrays/s, "
          f"{st.chunks} chunks; pixels equal to the exact frame {(d == 0).float().mean().item():.6f}, within 1 LSB {(d <= 1).float().mean().item():.6f}", flush=True)
