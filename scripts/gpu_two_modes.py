"""Render N frames of one workload in the one-launch pipeline, then N in the per-bounce pipeline (for rocprofv3 passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd import scenes
kind, w, h, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
hs = (scenes.instanced_scene(709, w, h) if kind == "inst" else scenes.mesh_scene(709, w, h)).build()
cam = hs.camera(0)
ds = rt.DeviceScene(hs, 0)
for flags in (0, rt.RT_FLAG_PER_BOUNCE_LAUNCHES):
    ds.set_option(rt.RT_OPT_FLAGS, flags)
    for _ in range(n):
        ds.render(cam, stats=rt.Stats())
ds.close()
