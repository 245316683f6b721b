"""Probe: mesh BVHs rebuilt on the device (binned SAH: "device"; linear BVH: "lbvh") against the host SAH rebuild (True) and the caller's trees traced
nearest-first (False): upload time, render time, equality.  RT_UPLOAD_TIMING=1 in the environment prints the phases and every tree's SAH cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
from raytracer_amd import scenes

def run(tag, spec, reps=8):
    hs = spec.build()
    cam = hs.camera(0)
    ref = None
    for mode in (False, True, "device", "lbvh"):
        ups = []
        ds = None
        for _ in range(3):
            if ds is not None: ds.close()
            t = time.perf_counter()
            ds = rt.DeviceScene(hs, 0, rebuild_mesh_bvh=mode)
            ups.append(time.perf_counter() - t)
        ds.set_option(rt.RT_OPT_FLAGS, rt.RT_FLAG_TIMING | rt.RT_FLAG_PER_BOUNCE_LAUNCHES | (rt.RT_FLAG_NEAR_FIRST if mode is False else 0))
        best = None
        rgb, hits = ds.render(cam, want_hits=True)
        for _ in range(reps):
            st = rt.Stats()
            ds.render(cam, stats=st)
            if best is None or st.ms_total < best.ms_total: best = st
        if ref is None: ref = (rgb, hits)
        rays = best.rays_primary + best.rays_secondary + best.rays_shadow
        print(f"{tag:16s} {str(mode):7s} upload {min(ups)*1e3:8.1f} ms | frame {best.ms_total:7.3f} ms {rays / best.ms_total / 1e3:7.1f} Mrays/s closest {best.ms_closest:.3f} shadow {best.ms_shadow:.3f} | "
              f"pixels != near-first exact {int((rgb != ref[0]).any(-1).sum())} hits != {int((hits != ref[1]).any(-1).sum())}", flush=True)
        ds.close()

which = sys.argv[1:] or ["small", "mesh"]
if "small" in which:
    run("spheres", scenes.spheres_scene(200, 150), 3)
    run("mesh g96", scenes.mesh_scene(grid=96, width=640, height=360), 3)
    run("inst g64", scenes.instanced_scene(grid=64, width=480, height=270), 3)
if "mesh" in which: run("mesh1m 1080p", scenes.mesh_scene(709, 1920, 1080))
if "inst" in which: run("inst10 1080p", scenes.instanced_scene(709, 1920, 1080), 4)
