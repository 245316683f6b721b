#!/usr/bin/env python3
"""Where does RT_FLAG_NEAR_FIRST differ from the exact mode?  gpu_near_probe.py golden:<name> | mesh:g:w:h | inst:g:w:h"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import numpy as np
import raytracer_amd as rt
from raytracer_amd import scenes
import util

for spec in sys.argv[1:]:
    kind, *a = spec.split(":")
    if kind == "golden":
        hs = util.golden_scene(a[0])
    elif kind == "mesh":
        hs = scenes.mesh_scene(int(a[0]), int(a[1]), int(a[2])).build()
    else:
        hs = scenes.instanced_scene(int(a[0]), int(a[1]), int(a[2])).build()
    cam = hs.camera(0)
    ds = rt.DeviceScene(hs, 0)
    rgb0, h0 = ds.render(cam, want_hits=True)
    if os.environ.get("RT_REBUILD") == "1":      # compare the exact mode with the library's own SAH trees instead
        ds.close()
        ds = rt.DeviceScene(hs, 0, rebuild_mesh_bvh=True)
    else:
        ds.set_option(rt.RT_OPT_FLAGS, rt.RT_FLAG_NEAR_FIRST)
    rgb1, h1 = ds.render(cam, want_hits=True)
    ds.close()
    a0, a1 = h0.reshape(-1, 2), h1.reshape(-1, 2)
    diff = (a0 != a1).any(axis=1)
    obj_diff = a0[:, 0] != a1[:, 0]
    cd = (rgb0.reshape(-1, 3) != rgb1.reshape(-1, 3)).any(axis=1)
    print(f"{spec}: {int(diff.sum())} of {diff.size} primary hits differ ({int(obj_diff.sum())} in the object, the rest in the face only); "
          f"{int(cd.sum())} colours differ, {int((cd & ~diff).sum())} of them with equal primary hits")
    idx = np.flatnonzero(diff)[:12]
    for i in idx:
        print(f"   pixel {i % cam.image_width},{i // cam.image_width}: exact {a0[i, 0]:#x}/{a0[i, 1]}  near {a1[i, 0]:#x}/{a1[i, 1]}")
