#!/usr/bin/env python3
"""inst10 (depth 6): all generations in one launch (RT_FLAG_ONE_LAUNCH) against the per-bounce pipeline, by render size —
the 1080p frame and one rank's share of the 8K frame at N = 8, 4, 2."""
import hashlib, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from raytracer_amd import DeviceScene, Stats, RT_OPT_FLAGS, RT_FLAG_TIMING, RT_FLAG_ONE_LAUNCH, RT_FLAG_PER_BOUNCE_LAUNCHES, scenes
for name, spec, parts in (("inst10 1080p", scenes.instanced_scene(709, 1920, 1080), 1), ("inst10 8K share 1/8", scenes.instanced_scene(709, 7680, 4320), 8),
                          ("inst10 8K share 1/4", scenes.instanced_scene(709, 7680, 4320), 4), ("inst10 8K share 1/2", scenes.instanced_scene(709, 7680, 4320), 2)):
    hs = spec.build()
    cam = hs.camera(0)
    ds = DeviceScene(hs, 0)
    W, H = cam.image_width, cam.image_height
    fb = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    for label, fl in (("per bounce", RT_FLAG_PER_BOUNCE_LAUNCHES), ("one launch", RT_FLAG_ONE_LAUNCH)):
        ds.set_option(RT_OPT_FLAGS, RT_FLAG_TIMING | fl)
        best = None
        for _ in range(5):
            st = Stats()
            ds.render_rows_device(cam, 0, 0, W, H, parts, 0, fb.data_ptr(), W * 3, stats=st)
            if best is None or st.ms_total < best.ms_total: best = st
        h = hashlib.sha256(fb.cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"{name:22s} {label}: {best.ms_total:8.3f} ms | closest {best.ms_closest:.3f} shadow {best.ms_shadow:.3f} shade {best.ms_shade:.3f} combine {best.ms_combine:.3f} chunks {best.chunks} | {h}", flush=True)
    ds.close()
