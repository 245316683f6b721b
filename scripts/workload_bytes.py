#!/usr/bin/env python3
"""Algorithmic bytes per ray of the bench workloads (SURVEY.md §8(d) formula),
counted by the CPU restatement in the reference's traversal order.  Writes
profiles/workload_bytes.json, which bench.py reads for roofline.achieved."""
import json, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from raytracer_amd import scenes
from oracle import binding as ob

out = {}
for name, sc in [("mesh1m_1920x1080", scenes.mesh_scene(709, 1920, 1080)),
                 ("inst10_1920x1080", scenes.instanced_scene(709, 1920, 1080))]:
    hs = sc.build()
    cam = hs.camera(0)
    t = time.time()
    _, _, c = ob.render(hs, cam, want_rgb=True)
    dt = time.time() - t
    out[name] = {
        "closest_rays": c.c_rays, "shadow_rays": c.s_rays,
        "closest_bytes_per_ray": c.closest_bytes() / max(c.c_rays, 1),
        "shadow_bytes_per_ray": c.shadow_bytes() / max(c.s_rays, 1),
        "closest": {k: getattr(c, "c_" + k) for k in ("top_nodes", "mesh_nodes", "tri_tests", "sphere_tests", "plane_tests", "inst_entries")},
        "shadow": {k: getattr(c, "s_" + k) for k in ("top_nodes", "mesh_nodes", "tri_tests", "sphere_tests", "plane_tests", "inst_entries")},
        "max_stack": c.max_stack, "oracle_seconds": dt,
        "formula": "32*(N_top+N_mesh) + 36*T_tri + 16*T_sph + 16*T_pln + 96*E_inst + 48 per ray (SURVEY.md 8d)",
    }
    print(name, json.dumps(out[name]), flush=True)
with open(os.path.join(ROOT, "profiles", "workload_bytes.json"), "w") as f:
    json.dump(out, f, indent=1)
