#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from raytracer_amd import DeviceScene, Stats, scenes
hs = scenes.mesh_scene(grid=96, width=640, height=360).build()
ds = DeviceScene(hs, 0)
st = Stats()
try:
    rgb, hits = ds.render(hs.camera(0), want_hits=True, stats=st)
    print("rendered", rgb.mean(), st.rays_primary, st.rays_shadow, st.ms_closest, st.ms_shadow)
except Exception as e:
    print("ERROR", e)
