"""Frames of an animation over the same meshes: rt_update_scene per frame against rt_upload_scene per
frame, on the 10-instance scene (1M-triangle mesh).  Host-side scene building (the reference's parser + BVH
builders, out of scope) is not timed: the descriptions are built beforehand."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import DeviceScene, scenes

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 709
W, H = 1920, 1080
frames = [scenes.instanced_scene(grid=grid, width=W, height=H, frame=f).build() for f in (0, 1, 2, 1, 2, 1, 2)]
ds = DeviceScene(frames[0], 0)
ref = []
t_up = []
for hs in frames[1:]:
    t0 = time.perf_counter()
    d2 = DeviceScene(hs, 0)
    rgb, _ = d2.render(hs.camera(0))
    t_up.append(time.perf_counter() - t0)
    d2.close()
    ref.append(rgb)
t_upd, t_call = [], []
for hs, want in zip(frames[1:], ref):
    t0 = time.perf_counter()
    ds.update(hs)
    t1 = time.perf_counter()
    rgb, _ = ds.render(hs.camera(0))
    t_upd.append(time.perf_counter() - t0)
    t_call.append(t1 - t0)
    assert (rgb == want).all()
print(f"grid {grid}: per frame (host output, {W}x{H}): upload+render {1e3 * np.median(t_up):.1f} ms, update+render {1e3 * np.median(t_upd):.1f} ms "
      f"(rt_update_scene alone {1e3 * np.median(t_call):.2f} ms); frames identical", flush=True)
