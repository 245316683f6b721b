"""Diagnostic (build -DRT_TIMELINE=2): the rays of the instanced frame's mixed launches that live longest."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd import scenes
hs = scenes.instanced_scene(709, 1920, 1080).build()
cam = hs.camera(0)
ds = rt.DeviceScene(hs, 0)
ds.set_option(rt.RT_OPT_FLAGS, rt.RT_FLAG_TIMING | rt.RT_FLAG_PER_BOUNCE_LAUNCHES)
for _ in range(2):
    st = rt.Stats()
    ds.render(cam, stats=st)
    print("frame", st.ms_total, "ms", flush=True)
ds.close()
