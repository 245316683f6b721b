#!/usr/bin/env python3
"""Quick GPU sanity run: HIP path vs the committed reference goldens."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import numpy as np
from raytracer_amd import DeviceScene, Stats, RT_OPT_FLAGS, RT_FLAG_TIMING
import util

names = sys.argv[1:] or ["hw1_simple", "hw1_two_spheres", "hw1_spheres_mirror", "hw1_cornellbox_recursive", "hw1_spheres_with_plane",
                         "hw2_mirror_room", "hw2_metal_glass_plates", "hw2_two_berserkers", "hw2_ellipsoids", "hw2_simple_transform",
                         "hw3_spheres_dof", "hw3_cornellbox_boxes_dynamic"]
for n in names:
    hs = util.golden_scene(n)
    cam = hs.camera(0)
    ds = DeviceScene(hs, 0)
    ds.set_option(RT_OPT_FLAGS, RT_FLAG_TIMING)
    st = Stats()
    t = time.time()
    rgb, hits = ds.render(cam, want_hits=True, stats=st)
    dt = time.time() - t
    g = util.golden_image(n)
    ok = util.hit_ids_match(hits, util.golden_hits(n))
    s = util.image_stats(rgb, g)
    rays = st.rays_primary + st.rays_secondary + st.rays_shadow
    print(f"{n:32s} img exact={s['exact']:.6f} within1={s['within1']:.6f} max={s['maxdiff']} hit-id={ok.mean():.6f} "
          f"rays={rays} ms_total={st.ms_total:.2f} closest={st.ms_closest:.2f} shadow={st.ms_shadow:.2f} wall={dt:.2f}s", flush=True)
    ds.close()
