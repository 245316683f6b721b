import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from raytracer_amd import DeviceScene, Stats, RT_OPT_FLAGS, RT_FLAG_TIMING, scenes
for which in ("both", "first", "second"):
    for (w, h) in ((1920, 1080), (3840, 2160)):
        s = scenes.mesh_scene(709, w, h)
        if which == "first": s.point_lights = s.point_lights[:1]
        if which == "second": s.point_lights = s.point_lights[1:]
        hs = s.build(); cam = hs.camera(0)
        ds = DeviceScene(hs, 0); ds.set_option(RT_OPT_FLAGS, RT_FLAG_TIMING)
        fb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
        best = None
        for it in range(4):
            st = Stats(); ds.render_device(cam, 0, 0, w, h, fb.data_ptr(), w * 3, stats=st)
            if best is None or st.ms_shadow < best.ms_shadow: best = st
        print(f"lights={which:6s} {w}x{h}: shadow rays {best.rays_shadow/1e6:.2f} M in {best.ms_shadow:.3f} ms = {best.rays_shadow/best.ms_shadow/1e3:.0f} Mrays/s", flush=True)
        ds.close()
