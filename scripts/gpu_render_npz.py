#!/usr/bin/env python3
"""Render scenes with whichever HIP library RT_HIP_SO names and save image + hit ids:
   gpu_render_npz.py out.npz golden:<name> | mesh:<grid>:<w>:<h> | inst:<grid>:<w>:<h> ...   (used by the GPU tests)"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import numpy as np
from raytracer_amd import DeviceScene, scenes
import util

out = {}
for spec in sys.argv[2:]:
    kind, *a = spec.split(":")
    if kind == "golden":
        hs = util.golden_scene(a[0])
    elif kind == "mesh":
        hs = scenes.mesh_scene(int(a[0]), int(a[1]), int(a[2])).build()
    elif kind == "glossy":
        hs = scenes.glossy_area_scene(int(a[0]), int(a[1]), int(a[2]), int(a[3])).build()
    else:
        hs = scenes.instanced_scene(int(a[0]), int(a[1]), int(a[2])).build()
    ds = DeviceScene(hs, 0)
    rgb, hits = ds.render(hs.camera(0), want_hits=True)
    ds.close()
    out[spec + "/rgb"], out[spec + "/hits"] = rgb, hits
np.savez(sys.argv[1], **out)
print("ok", len(sys.argv) - 2)
