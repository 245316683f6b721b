#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import numpy as np
from raytracer_amd import DeviceScene, Stats
from oracle import binding as ob
import util
n = sys.argv[1]
hs = util.golden_scene(n); cam = hs.camera(0)
for depth in range(0, 7):
    hs.set_max_depth(depth)
    ds = DeviceScene(hs, 0); st = Stats()
    rgb, _ = ds.render(cam, stats=st)
    org, _, ctr = ob.render(hs, cam)
    d = np.abs(rgb.astype(int) - org.astype(int)).max(-1)
    print(n, "depth", depth, util.image_stats(rgb, org), "gpu rays", st.rays_primary, st.rays_secondary, st.rays_shadow,
          "oracle", ctr.rays_primary, ctr.rays_secondary, ctr.s_rays, flush=True)
    np.savez_compressed(f"gpurun_out/{n}.d{depth}.npz", gpu=rgb, ora=org)
    ds.close()
