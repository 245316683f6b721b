#!/usr/bin/env python3
import sys, os, ctypes as C
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import numpy as np
from raytracer_amd import DeviceScene, Stats, hip_lib
from oracle import binding as ob
import util
n = sys.argv[1]
hs = util.golden_scene(n); cam = hs.camera(0); hs.set_max_depth(0)
ds = DeviceScene(hs, 0)
rgb, _ = ds.render(cam)
org, _, _ = ob.render(hs, cam)
w, h = cam.image_width, cam.image_height
L = hip_lib(); L.rt_debug_fetch.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p]
tpr = (w + 7)//8; nrec = tpr*((h+7)//8)*64
def fetch(which, count, dt, width):
    a = np.zeros((count, width), dt); assert L.rt_debug_fetch(ds._h, which, 0, count, a.ctypes.data) == 0; return a
ray0 = fetch(0, nrec, np.float32, 4); ray1 = fetch(1, nrec, np.float32, 4); hit0 = fetch(2, nrec, np.uint32, 4)
nrm = fetch(3, nrec, np.float32, 4); link = fetch(4, nrec, np.uint32, 4)
nl = 3
lit = fetch(6, nrec*nl, np.uint8, 1).ravel()
O = ob.lib(); O.rto_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
d = np.abs(rgb.astype(int)-org.astype(int)).max(-1)
ys, xs = np.nonzero(d > 0)
print("bad", len(ys))
def rec_of(x, y): return ((y//8)*tpr + x//8)*64 + (y%8)*8 + x%8
cnt = 0
for y, x in list(zip(ys, xs))[:12]:
    r = rec_of(x, y)
    out = np.zeros(28, np.float32); O.rto_probe(hs.desc_ptr, C.byref(cam), int(x), int(y), out.ctypes.data)
    g = np.concatenate([ray0[r,:3], ray1[r,:3], ray0[r,3:4]])
    t_g = hit0[r,0:1].view(np.float32)[0]
    sb = link[r,2]
    print(f"px {x},{y} rec {r} gpu rgb {rgb[y,x]} ora {org[y,x]}")
    print("  ray equal:", (g.view(np.uint32) == out[:7].view(np.uint32)).all(), " t gpu/ora bits", hex(hit0[r,0]), hex(out[8:9].view(np.uint32)[0]), "kind", hit0[r,1]>>28, out[9])
    print("  n equal:", (nrm[r,:3].view(np.uint32) == out[11:14].view(np.uint32)).all(), nrm[r,:3], out[11:14])
    K = int(link[r,3]) >> 8          # the record's slot of light l is sb + l*K (k_shade: light-major inside a block)
    print("  lit gpu", [int(lit[sb + l * K]) for l in range(nl)] if K else [], "ora", out[20:20+nl], "flags", link[r,3] & 0xFF)
