#!/usr/bin/env python3
"""Algorithmic bytes per ray of the bench workloads (SURVEY.md §8(d) formula),
counted by the CPU restatement in the reference's traversal order.  Writes
profiles/workload_bytes.json, which bench.py reads for roofline.achieved."""
import json, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from raytracer_amd import scenes
from oracle import binding as ob

# usage: workload_bytes.py [workload names...]   (no names = all; entries of other workloads are kept)
path = os.path.join(ROOT, "profiles", "workload_bytes.json")
out = json.load(open(path)) if os.path.exists(path) else {}
WORKLOADS = [("mesh1m_1920x1080", lambda: scenes.mesh_scene(709, 1920, 1080)),
             ("inst10_1920x1080", lambda: scenes.instanced_scene(709, 1920, 1080)),
             # BASELINE configs[4] / SURVEY 8d-M5: the frame bench.py --gpus N splits over N GPUs
             ("inst10_7680x4320", lambda: scenes.instanced_scene(709, 7680, 4320))]
for name, make in WORKLOADS:
    if sys.argv[1:] and name not in sys.argv[1:]:
        continue
    sc = make()
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
with open(path, "w") as f:
    json.dump(out, f, indent=1)
