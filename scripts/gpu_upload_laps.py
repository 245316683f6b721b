#!/usr/bin/env python3
"""Phases of one rt_upload_scene_ex of the 1 M-triangle bench scene (RT_UPLOAD_TIMING=1 prints them) per build mode, third upload of each.
usage: gpu_upload_laps.py [modes: exact host device lbvh]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_UPLOAD_TIMING"] = "1"
import raytracer_amd as rt
from raytracer_amd import scenes
hs = scenes.mesh_scene(709, 1920, 1080).build()
for mode in (sys.argv[1:] or ["exact", "host", "device", "lbvh"]):
    arg = {"exact": False, "host": True, "device": "device", "lbvh": "lbvh"}[mode]
    for k in range(3):
        sys.stderr.write(f"==== {mode} upload {k}\n"); sys.stderr.flush()
        t = time.perf_counter()
        ds = rt.DeviceScene(hs, 0, rebuild_mesh_bvh=arg)
        dt = time.perf_counter() - t
        sys.stderr.write(f"==== {mode} upload {k}: {1e3 * dt:.1f} ms\n"); sys.stderr.flush()
        ds.close()
