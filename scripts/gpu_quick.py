#!/usr/bin/env python3
"""Quick A/B of library builds (RT_HIP_SO=librt_hip_<variant>.so): per workload the best of a few frames, per-stage times
and a checksum of frame + hit ids + ray counts (equal checksums = identical results).  usage: gpu_quick.py [case substrings]"""
import hashlib, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from raytracer_amd import DeviceScene, Stats, RT_OPT_FLAGS, RT_FLAG_TIMING, scenes
cases = [
    ("mesh1m 1080p", lambda: scenes.mesh_scene(709, 1920, 1080)),
    ("inst10 1080p", lambda: scenes.instanced_scene(709, 1920, 1080)),
    ("mesh1m 4K", lambda: scenes.mesh_scene(709, 3840, 2160)),
    ("inst10 8K", lambda: scenes.instanced_scene(709, 7680, 4320)),
    ("glossy 1080p 16spp", lambda: scenes.glossy_area_scene(256, 1920, 1080, 16)),
]
sel = sys.argv[1:] or ["mesh1m 1080p", "inst10 1080p"]
iters = int(os.environ.get("RT_QUICK_ITERS", "8"))
flags = int(os.environ.get("RT_QUICK_FLAGS", "0"))
for name, mk in cases:
    if not any(s in name for s in sel):
        continue
    hs = mk().build()
    cam = hs.camera(0)
    ds = DeviceScene(hs, 0)
    ds.set_option(RT_OPT_FLAGS, RT_FLAG_TIMING | flags)
    W, H = cam.image_width, cam.image_height
    fb = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    hb = torch.zeros((H, W, 2), dtype=torch.int32, device="cuda")
    best = None
    tot = []
    for it in range(iters):
        st = Stats()
        ds.render_device(cam, 0, 0, W, H, fb.data_ptr(), W * 3, d_hits_ptr=hb.data_ptr(), stats=st)
        tot.append(st.ms_total)
        if best is None or st.ms_total < best.ms_total:
            best = st
    rays = best.rays_primary + best.rays_secondary + best.rays_shadow
    h = hashlib.sha256(fb.cpu().numpy().tobytes() + hb.cpu().numpy().tobytes() + repr((best.rays_primary, best.rays_secondary, best.rays_shadow)).encode()).hexdigest()[:12]
    tot.sort()
    print(f"{os.environ.get('RT_HIP_SO', 'librt_hip.so'):24s} {name:20s} best {best.ms_total:8.3f} med {tot[len(tot)//2]:8.3f} ms {rays/best.ms_total/1e3:8.1f} Mrays/s | closest {best.ms_closest:.3f} shadow {best.ms_shadow:.3f} shade {best.ms_shade:.3f} raygen {best.ms_raygen:.3f} | {h}", flush=True)
    ds.close()
