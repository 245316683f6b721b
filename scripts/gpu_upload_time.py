"""Time rt_upload_scene (host re-layout + copies) for the bench scene and an instanced scene."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import DeviceScene, scenes

for name, spec in (("mesh1m", scenes.mesh_scene(709, 1920, 1080)), ("inst10", scenes.instanced_scene(grid=709, width=1920, height=1080))):
    t0 = time.perf_counter(); hs = spec.build(); t1 = time.perf_counter()
    ds = DeviceScene(hs, 0); t2 = time.perf_counter()
    ds.close()
    ds = DeviceScene(hs, 0); t3 = time.perf_counter()
    print(f"{name}: host build {t1 - t0:.3f} s, first rt_upload_scene {t2 - t1:.3f} s, second {t3 - t2:.3f} s", flush=True)
    ds.close()
