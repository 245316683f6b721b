"""Probe: frame 1 of the 48-grid animation (328x184) — fresh upload vs update path vs oracle; where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
from raytracer_amd import scenes
from oracle import binding as ob

def show(tag, rgb, hits, o_rgb, o_hits):
    bad = np.argwhere((rgb != o_rgb).any(-1))
    badh = np.argwhere((hits != o_hits).any(-1))
    print(tag, "pixels differ:", len(bad), "hit ids differ:", len(badh), flush=True)
    for y, x in bad[:5]:
        print("   px", x, y, "gpu", rgb[y, x], "oracle", o_rgb[y, x], "hit gpu", [hex(v) for v in hits[y, x]], "oracle", [hex(v) for v in o_hits[y, x]])

for grid, w, h in ((48, 328, 184), (40, 192, 108), (64, 480, 270)):
    f0 = scenes.instanced_scene(grid=grid, width=w, height=h, frame=0).build()
    for frame in (1, 2):
        hs = scenes.instanced_scene(grid=grid, width=w, height=h, frame=frame).build()
        cam = hs.camera(0)
        o_rgb, o_hits, ctr = ob.render(hs, cam, want_hits=True)
        ds = rt.DeviceScene(hs, 0)
        st = rt.Stats()
        rgb, hits = ds.render(cam, want_hits=True, stats=st)
        show(f"grid {grid} frame {frame} fresh upload:", rgb, hits, o_rgb, o_hits)
        print("   rays gpu", st.rays_primary, st.rays_secondary, st.rays_shadow, "oracle", ctr.rays_primary, ctr.rays_secondary, ctr.s_rays)
        ds.close()
        ds = rt.DeviceScene(f0, 0)
        ds.render(f0.camera(0))
        ds.update(hs, verify_meshes=True)
        rgb2, hits2 = ds.render(cam, want_hits=True)
        show(f"grid {grid} frame {frame} after update:", rgb2, hits2, o_rgb, o_hits)
        ds.close()
