#!/usr/bin/env python3
"""Other BASELINE.json configurations as performance cases (not bench lines): one line each.
   RT_NEAR=1 in the environment selects the non-parity visiting order (RT_FLAG_NEAR_FIRST), RT_REBUILD=1 the
   library's own SAH trees (rt_upload_scene_ex, RT_UPLOAD_REBUILD_MESH_BVH)."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from raytracer_amd import DeviceScene, Stats, RT_OPT_FLAGS, RT_FLAG_TIMING, RT_FLAG_NEAR_FIRST, scenes
NEAR = RT_FLAG_NEAR_FIRST if os.environ.get("RT_NEAR") == "1" else 0
cases = [
    ("mesh1m 1920x1080 1spp depth0 (configs[1])", lambda: scenes.mesh_scene(709, 1920, 1080)),
    ("mesh1m 3840x2160 1spp depth0", lambda: scenes.mesh_scene(709, 3840, 2160)),
    ("inst10 1920x1080 1spp depth6 (configs[2])", lambda: scenes.instanced_scene(709, 1920, 1080)),
    ("inst10 3840x2160 16spp depth6 (configs[3] scaled)", lambda: scenes.instanced_scene(709, 3840, 2160, num_samples=16)),
    ("inst10 7680x4320 1spp depth6 (configs[4], one GPU)", lambda: scenes.instanced_scene(709, 7680, 4320)),
    ("glossy_area 1920x1080 16spp depth4 (8f-1)", lambda: scenes.glossy_area_scene(256, 1920, 1080, 16)),
]
sel = sys.argv[1:]
for name, mk in cases:
    if sel and not any(s in name for s in sel):
        continue
    hs = mk().build()
    cam = hs.camera(0)
    ds = DeviceScene(hs, 0, rebuild_mesh_bvh=os.environ.get("RT_REBUILD") == "1")
    ds.set_option(RT_OPT_FLAGS, RT_FLAG_TIMING | NEAR)
    import torch
    fb = torch.zeros((cam.image_height, cam.image_width, 3), dtype=torch.uint8, device="cuda")
    best = None
    for it in range(3):
        st = Stats()
        t = time.perf_counter()
        ds.render_device(cam, 0, 0, cam.image_width, cam.image_height, fb.data_ptr(), cam.image_width * 3, stats=st)
        dt = time.perf_counter() - t
        if best is None or st.ms_total < best[0].ms_total:
            best = (st, dt)
    st, dt = best
    rays = st.rays_primary + st.rays_secondary + st.rays_shadow
    print(f"{name:52s} rays={rays/1e6:9.1f}M  {st.ms_total:9.2f} ms  {rays/st.ms_total/1e3:8.1f} Mrays/s | raygen {st.ms_raygen:.2f} closest {st.ms_closest:.2f} "
          f"shade {st.ms_shade:.2f} shadow {st.ms_shadow:.2f} combine {st.ms_combine:.2f} resolve {st.ms_resolve:.2f} chunks {st.chunks}", flush=True)
    ds.close()
