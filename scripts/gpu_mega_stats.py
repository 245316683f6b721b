"""Stats-build probe (librt_hip_stats*.so via RT_HIP_SO): one frame of mesh1m 4K per pipeline; the library prints its step statistics on stderr."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd import scenes
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160)
hs = (scenes.instanced_scene(709, w, h) if sys.argv[1] == "inst" else scenes.mesh_scene(709, w, h)).build()
cam = hs.camera(0)
ds = rt.DeviceScene(hs, 0)
for name, flags in (("mega", 0), ("bounce", rt.RT_FLAG_PER_BOUNCE_LAUNCHES)):
    ds.set_option(rt.RT_OPT_FLAGS, flags | rt.RT_FLAG_TIMING)
    ds.render(cam, stats=rt.Stats())
    sys.stderr.flush()
    print(f"===== {name}", file=sys.stderr, flush=True)
    st = rt.Stats()
    ds.render(cam, stats=st)
    print(f"===== {name}: {st.ms_total:.3f} ms", file=sys.stderr, flush=True)
ds.close()
