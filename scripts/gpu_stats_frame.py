"""Stats-build probe (librt_hip_stats*.so via RT_HIP_SO): one frame, per-bounce pipeline; the library prints its step statistics on stderr.
usage: RT_HIP_SO=... gpu_stats_frame.py mesh|inst W H"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd import scenes
w, h = int(sys.argv[2]), int(sys.argv[3])
hs = (scenes.instanced_scene(709, w, h) if sys.argv[1] == "inst" else scenes.mesh_scene(709, w, h)).build()
cam = hs.camera(0)
ds = rt.DeviceScene(hs, 0)
ds.set_option(rt.RT_OPT_FLAGS, rt.RT_FLAG_PER_BOUNCE_LAUNCHES | rt.RT_FLAG_TIMING)
ds.render(cam, stats=rt.Stats())
sys.stderr.flush()
print("===== measured frame", file=sys.stderr, flush=True)
st = rt.Stats()
ds.render(cam, stats=st)
print(f"===== {st.ms_total:.3f} ms closest {st.ms_closest:.3f} shadow {st.ms_shadow:.3f}", file=sys.stderr, flush=True)
ds.close()
