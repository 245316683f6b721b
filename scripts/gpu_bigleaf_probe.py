#!/usr/bin/env python3
"""The dragon_new pathology as a performance case (blog/hw2/episode2.md:233-248: 13 min 14 s on the author's CPU):
one mesh = a displaced sphere of ~200 k triangles plus one huge ground triangle.  The huge triangle stretches the
root box so that every centroid lies left of its midpoint, SubdivideMesh (raytracer.cpp:296-339) gives up at the
root, and the whole mesh is ONE leaf.  Exact mode (the reference's tree: every ray tests every triangle, 64 at a
time by the whole wave) against rt_upload_scene_ex(RT_UPLOAD_REBUILD_MESH_BVH)."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import raytracer_amd as rt
from raytracer_amd import scenes

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 317
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 360)
s = scenes.mesh_scene(grid, w, h, name="one_leaf")
verts, faces = scenes.displaced_sphere(grid)
ground = np.array([(-6, -1.15, -6), (-6, -1.15, 400), (400, -1.15, -6)], np.float32)
s.meshes = [dict(verts=np.concatenate([verts, ground]), faces=np.concatenate([faces, np.array([[len(verts), len(verts) + 1, len(verts) + 2]], faces.dtype)]), material=1)]
hs = s.build()
cam = hs.camera(0)
out = {}
for mode in ("exact", "rebuilt", "device"):
    t = time.perf_counter()
    ds = rt.DeviceScene(hs, 0, rebuild_mesh_bvh=(True if mode == "rebuilt" else ("device" if mode == "device" else False)))
    up = time.perf_counter() - t
    ds.set_option(rt.RT_OPT_FLAGS, rt.RT_FLAG_TIMING)
    best = None
    for it in range(2):
        st = rt.Stats()
        rgb, hits = ds.render(cam, want_hits=True, stats=st)
        if best is None or st.ms_total < best.ms_total:
            best = st
    ds.close()
    rays = best.rays_primary + best.rays_secondary + best.rays_shadow
    out[mode] = (rgb, hits)
    print(f"{mode:8s} {len(faces) + 1} triangles, {w}x{h}: upload {up*1e3:8.1f} ms, frame {best.ms_total:10.2f} ms, {rays/best.ms_total/1e3:9.2f} Mrays/s "
          f"(closest {best.ms_closest:.2f} shadow {best.ms_shadow:.2f})", flush=True)
same = (out["exact"][1].reshape(-1, 2) == out["rebuilt"][1].reshape(-1, 2)).all(axis=1)
d = np.abs(out["exact"][0].astype(np.int16) - out["rebuilt"][0].astype(np.int16)).max(axis=-1)
print(f"hit ids equal {same.mean():.6f}, colours equal {(d == 0).mean():.6f}, within 1 LSB {(d <= 1).mean():.6f}")
