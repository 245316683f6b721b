#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import numpy as np
from raytracer_amd import DeviceScene
import util
for n in sys.argv[1:]:
    hs = util.golden_scene(n); cam = hs.camera(0); ds = DeviceScene(hs, 0)
    rgb, hits = ds.render(cam, want_hits=True)
    np.savez_compressed(f"gpurun_out/{n}.gpu.npz", rgb=rgb, hits=hits)
    print(n, util.image_stats(rgb, util.golden_image(n)))
