"""One library build (RT_HIP_SO) on the two 1080p workloads, per-bounce pipeline: best-of-N stage times.
usage: RT_HIP_SO=librt_hip_x.so gpu_variant_bench.py [mesh] [inst] [mesh4k]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd import scenes

def run(tag, spec, reps):
    hs = spec.build()
    cam = hs.camera(0)
    ds = rt.DeviceScene(hs, 0)
    ds.set_option(rt.RT_OPT_FLAGS, rt.RT_FLAG_PER_BOUNCE_LAUNCHES | rt.RT_FLAG_TIMING)
    best = None
    for _ in range(reps + 2):
        st = rt.Stats()
        ds.render(cam, stats=st)
        if best is None or st.ms_total < best.ms_total:
            best = st
    rays = best.rays_primary + best.rays_secondary + best.rays_shadow
    print(f"{os.environ.get('RT_HIP_SO', 'librt_hip.so'):24s} {tag:14s} total {best.ms_total:7.3f} ms {rays / best.ms_total / 1e3:7.1f} Mrays/s | closest {best.ms_closest:.3f} shadow {best.ms_shadow:.3f} shade {best.ms_shade:.3f}", flush=True)
    ds.close()

which = sys.argv[1:] or ["mesh", "inst"]
if "mesh" in which: run("mesh1m 1080p", scenes.mesh_scene(709, 1920, 1080), 10)
if "inst" in which: run("inst10 1080p", scenes.instanced_scene(709, 1920, 1080), 5)
if "mesh4k" in which: run("mesh1m 4K", scenes.mesh_scene(709, 3840, 2160), 5)
