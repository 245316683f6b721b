"""Probe: per-bounce pipeline (with end-game consolidation unless RT_NO_PARK=1) against the oracle on a small scene: where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
from raytracer_amd import scenes
from oracle import binding as ob
for tag, spec in (("spheres", scenes.spheres_scene(200, 150)), ("mesh g96", scenes.mesh_scene(grid=96, width=640, height=360)), ("inst g64", scenes.instanced_scene(grid=64, width=480, height=270))):
    hs = spec.build()
    cam = hs.camera(0)
    o_rgb, o_hits, ctr = ob.render(hs, cam, want_hits=True)
    ds = rt.DeviceScene(hs, 0)
    ds.set_option(rt.RT_OPT_FLAGS, rt.RT_FLAG_PER_BOUNCE_LAUNCHES | rt.RT_FLAG_TIMING)
    for rep in range(3):
        st = rt.Stats()
        rgb, hits = ds.render(cam, want_hits=True, stats=st)
        bad = (rgb != o_rgb).any(-1)
        badh = (hits != o_hits).any(-1)
        print(f"{tag} rep {rep}: pixels differ {int(bad.sum())} hits differ {int(badh.sum())} rays {st.rays_primary} {st.rays_secondary} {st.rays_shadow} oracle {ctr.rays_primary} {ctr.rays_secondary} {ctr.s_rays}", flush=True)
        ys, xs = np.nonzero(bad)
        for y, x in list(zip(ys, xs))[:6]:
            print(f"    ({x},{y}) gpu {rgb[y, x]} oracle {o_rgb[y, x]} hit gpu {hits[y, x]} oracle {o_hits[y, x]}")
    ds.close()
