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
def many_spheres(n_side, width=1920, height=1080):
    """A large top level: n_side x n_side spheres on a plane (top BVH of ~2 n^2 nodes), two point lights, a mirror floor."""
    import numpy as np
    from raytracer_amd import ffi
    s = scenes.SynthScene(f"spheres{n_side}")
    s.max_depth = 2
    verts = [(-40.0, -1.0, -40.0)]
    rng = np.random.default_rng(7)
    for i in range(n_side):
        for j in range(n_side):
            verts.append((float((i - n_side / 2) * 1.0 + rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.3, 0.3)), float((j - n_side / 2) * 1.0 + rng.uniform(-0.2, 0.2))))
    s.vertices = verts
    s.materials = [dict(ambient=(1, 1, 1), diffuse=(0.8, 0.4, 0.3), specular=(1, 1, 1), phong=30.0),
                   dict(type=ffi.RT_MAT_MIRROR, ambient=(0.1, 0.1, 0.1), diffuse=(0.3, 0.3, 0.3), specular=(1, 1, 1), mirror=(0.6, 0.6, 0.6), phong=80.0)]
    s.point_lights = [((10, 30, 20), (400000, 400000, 400000)), ((-20, 15, 5), (150000, 150000, 200000))]
    s.spheres = [dict(center=k + 2, radius=0.35, material=1 + (k % 7 == 0)) for k in range(n_side * n_side)]
    s.planes = [dict(point=1, normal=(0, 1, 0), material=1)]
    d = n_side * 0.55
    s.camera = dict(position=(0.0, d * 0.6, d), gaze_point=(0, 0, 0), up=(0, 1, 0), fovy=45.0, near=1.0, width=width, height=height, num_samples=1)
    return s
cases += [("spheres 30x30", lambda: many_spheres(30)), ("spheres 200x200", lambda: many_spheres(200))]
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
