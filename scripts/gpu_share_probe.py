"""Probe: one rank's share of the 8K inst10 frame (configs[4]) for N = 1, 2, 4, 8 interleaved row parts on ONE GPU:
time per share against the whole frame -> the strong-scaling efficiency the trace launches allow (no gather here)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
W, H = 7680, 4320
hs = scenes.instanced_scene(709, W, H).build()
cam = hs.camera(0)
ds = rt.DeviceScene(hs, 0)
flags = int(os.environ.get("RT_PROBE_FLAGS", "0"))
ds.set_option(rt.RT_OPT_FLAGS, rt.RT_FLAG_TIMING | flags)
fb = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
base = None
for n in (1, 2, 4, 8):
    best = None
    for _ in range(3):
        st = rt.Stats()
        ds.render_rows_device(cam, 0, 0, W, H, n, 0, fb.data_ptr(), W * 3, stats=st)
        if best is None or st.ms_total < best.ms_total: best = st
    rays = best.rays_primary + best.rays_secondary + best.rays_shadow
    if base is None: base = best.ms_total
    print(f"share 1/{n}: {best.ms_total:8.2f} ms  {rays/1e6:7.1f} M rays  {rays/best.ms_total/1e3:7.1f} Mrays/s on this GPU | speed-up if all {n} ranks took this long: {base/best.ms_total:5.2f} "
          f"(efficiency {base/best.ms_total/n:.2f}) | closest {best.ms_closest:.2f} shadow {best.ms_shadow:.2f} shade {best.ms_shade:.2f} launches {best.launches_closest}+{best.launches_shadow}", flush=True)
ds.close()
