"""rt_render_tile with HOST output (PCIe-inclusive) against rt_render_tile_device on the bench workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from raytracer_amd import DeviceScene, scenes
from raytracer_amd.ffi import hip_lib, _check
W, H = 1920, 1080
hs = scenes.mesh_scene(709, W, H).build()
cam = hs.camera(0)
ds = DeviceScene(hs, 0)
rgb = np.zeros((H, W, 3), np.uint8)
fb = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
L = hip_lib()
def host():
    _check(L.rt_render_tile(ds._h, C.byref(cam), 0, 0, W, H, rgb.ctypes.data, W * 3, None, None))
def dev():
    ds.render_device(cam, 0, 0, W, H, fb.data_ptr(), W * 3)
    torch.cuda.synchronize()
for name, f in (("rt_render_tile (host rgb_out, pageable)", host), ("rt_render_tile_device", dev)):
    for _ in range(3): f()
    t = time.perf_counter()
    for _ in range(20): f()
    dt = (time.perf_counter() - t) / 20
    print(f"{name}: {dt * 1e3:.3f} ms/frame, {5071806 / dt / 1e6:.0f} Mrays/s", flush=True)
assert (torch.from_numpy(rgb).cuda() == fb).all()
