#!/usr/bin/env python3
"""npz_diff.py a.npz b.npz: per entry of two gpu_render_npz.py outputs, the fraction of pixels whose hit ids agree and
   whose colours agree exactly / within 1 LSB (the north_star's tolerances are 99.99 % and 99.9 %)."""
import sys
import numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in a.files:
    x, y = a[k], b[k]
    if k.endswith("/hits"):
        same = (x.reshape(-1, 2) == y.reshape(-1, 2)).all(axis=1)
        print(f"{k:34s} hit ids equal {same.mean()*100:.5f} %  ({int((~same).sum())} of {same.size} differ)")
    else:
        d = np.abs(x.astype(np.int16) - y.astype(np.int16)).reshape(-1, 3).max(axis=1)
        print(f"{k:34s} byte-exact {(d == 0).mean()*100:.5f} %  within 1 LSB {(d <= 1).mean()*100:.5f} %  max diff {int(d.max())}")
